"""Device -> host copies of one 4K RGB frame (24.9 MB) into the kinds of buffer a plugin caller may hold, and where
those buffers' pages lie (NUMA node by move_pages).  The plugin's decode_image times moved by +-0.4 ms from process to
process with nothing changed but the allocation history (profiles/r5_wide_packs.md): this probe looks for the cause.
Usage: python tools/d2h_probe.py"""
import ctypes as C, mmap, os, time, glob
import numpy as np

hip = C.CDLL("libamdhip64.so")
libc = C.CDLL(None, use_errno=True)
libc.mmap.restype = C.c_void_p
libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
libc.munmap.argtypes = [C.c_void_p, C.c_size_t]
libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
libc.syscall.restype = C.c_long
N = 3840 * 2160 * 3


def ok(e):
    assert e == 0, e


def nodes_of(ptr, nbytes):
    """{node: pages} of the buffer (move_pages with no target nodes reports where each page is)."""
    n = (nbytes + 4095) // 4096
    pages = (C.c_void_p * n)(*[ptr + 4096 * i for i in range(n)])
    status = (C.c_int * n)()
    r = libc.syscall(279, 0, C.c_ulong(n), pages, None, status, 0)
    if r != 0:
        return {"?": C.get_errno()}
    out = {}
    for s in status:
        out[s] = out.get(s, 0) + 1
    return out


def time_copy(dst, d_src, stream, reps=20):
    ok(hip.hipMemcpyAsync(C.c_void_p(dst), d_src, C.c_size_t(N), 2, stream)); ok(hip.hipStreamSynchronize(stream))
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ok(hip.hipMemcpyAsync(C.c_void_p(dst), d_src, C.c_size_t(N), 2, stream))
        ok(hip.hipStreamSynchronize(stream))
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[0] * 1e3, ts[len(ts) // 2] * 1e3


def anon(nbytes, huge=False, align=4096):
    raw = libc.mmap(None, nbytes + align, 3, 0x22, -1, 0)             # PROT_READ|WRITE, MAP_PRIVATE|MAP_ANONYMOUS
    p = (raw + align - 1) // align * align
    if huge:
        libc.madvise(C.c_void_p(p), nbytes, 14)                         # MADV_HUGEPAGE
    return p


def touch(p, nbytes):
    C.memset(C.c_void_p(p), 1, nbytes)


def main():
    ok(hip.hipSetDevice(0))
    d_src = C.c_void_p(); ok(hip.hipMalloc(C.byref(d_src), C.c_size_t(N)))
    stream = C.c_void_p(); ok(hip.hipStreamCreateWithFlags(C.byref(stream), 1))
    cpus = sorted(os.sched_getaffinity(0))
    print("affinity: %d CPUs %s...%s" % (len(cpus), cpus[:4], cpus[-4:]))
    for nd in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        print(os.path.basename(nd), "cpus", open(nd + "/cpulist").read().strip())
    for dev in sorted(glob.glob("/sys/class/drm/card*/device/numa_node")):
        print(dev, open(dev).read().strip())
    try:
        print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
    except OSError:
        pass
    node_cpus = {}
    for nd in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        lst = []
        for part in open(nd + "/cpulist").read().strip().split(","):
            if not part:
                continue
            a, _, b = part.partition("-")
            lst += list(range(int(a), int(b or a) + 1))
        node_cpus[int(os.path.basename(nd)[4:])] = [c for c in lst if c in cpus]
    print("granted CPUs by node:", {k: len(v) for k, v in node_cpus.items()})

    def report(name, p):
        best, med = time_copy(p, d_src, stream)
        print("%-58s best %.3f ms median %.3f ms (%.1f GB/s)  pages by node %s" % (name, best, med, N / best / 1e6, nodes_of(p, N)), flush=True)

    # pinned by HIP
    h = C.c_void_p(); ok(hip.hipHostMalloc(C.byref(h), C.c_size_t(N), 0))
    report("hipHostMalloc", h.value)
    for node, cl in node_cpus.items():
        if not cl:
            continue
        os.sched_setaffinity(0, cl[:1])
        time.sleep(0.01)
        for huge in (False, True):
            p = anon(N, huge=huge, align=2 << 20 if huge else 4096)
            touch(p, N)
            report("anonymous pages touched on node %d%s, copy from there" % (node, ", MADV_HUGEPAGE 2 MB aligned" if huge else ""), p)
            os.sched_setaffinity(0, cpus)
            report("  the same buffer, thread free to move", p)
            ok(hip.hipHostRegister(C.c_void_p(p), C.c_size_t(N), 0))
            report("  the same buffer registered (hipHostRegister)", p)
            ok(hip.hipHostUnregister(C.c_void_p(p)))
            os.sched_setaffinity(0, cl[:1])
    os.sched_setaffinity(0, cpus)
    # what malloc gives (numpy) — several in a row, as a process with some history would
    keep = []
    for k in range(6):
        a = np.empty(N + (k * 12345 if k % 2 else 0), dtype=np.uint8)
        a[:] = 1
        keep.append(a)
        report("numpy buffer %d (address %% 2 MB = %d KB)" % (k, (a.ctypes.data % (2 << 20)) >> 10), a.ctypes.data)


if __name__ == "__main__":
    main()
