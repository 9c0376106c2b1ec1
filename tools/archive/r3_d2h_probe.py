"""D2H into caller memory: hipHostMalloc'ed, hipHostRegister'ed (malloc'ed, 4 KB pages), registered + 2 MB aligned
with MADV_HUGEPAGE; 1 / 2 / 4 streams."""
import ctypes as C, mmap, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from jpeg_gpu_amd import lib
N = 3840 * 2160 * 3
dev = lib.DeviceBuffer(N)
streams = [lib.L.jga_stream_create() for _ in range(4)]
libc = C.CDLL(None)


def run(ptr, label):
    for ns in (1, 2, 4):
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            step = (N // ns + 4095) // 4096 * 4096
            for i in range(ns):
                o = i * step
                n = min(step, N - o)
                lib.check(lib.L.jga_memcpy_d2h(ptr + o, dev.ptr + o, n, streams[i]))
            for i in range(ns):
                lib.check(lib.L.jga_stream_sync(streams[i]))
            best = min(best, time.perf_counter() - t0)
        print("%-44s %d stream(s): %.3f ms = %.1f GB/s" % (label, ns, best * 1e3, N / best / 1e9))


p = lib.L.jga_host_malloc_pinned(N)
run(p, "hipHostMalloc")
a = np.empty(N + 4096, np.uint8); a[:] = 1
q = (a.ctypes.data + 4095) // 4096 * 4096
lib.check(lib.L.jga_host_register(q, N))
run(q, "malloc + hipHostRegister")
lib.L.jga_host_unregister(q)
m = mmap.mmap(-1, N + (4 << 20))
base = C.addressof(C.c_char.from_buffer(m))
h = (base + (2 << 20) - 1) // (2 << 20) * (2 << 20)
libc.madvise(C.c_void_p(h), C.c_size_t((N + (2 << 20) - 1) // (2 << 20) * (2 << 20)), 14)   # MADV_HUGEPAGE
C.memset(h, 1, N)
lib.check(lib.L.jga_host_register(h, N))
run(h, "mmap 2 MB aligned + MADV_HUGEPAGE + register")
