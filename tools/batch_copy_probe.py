"""hipMemcpyBatchAsync against a loop of hipMemcpyAsync: 16 files of 0.78 MB (a group of config 4's shard) from pinned
and from registered ordinary memory into one device buffer, one stream.  Usage: python tools/batch_copy_probe.py"""
import ctypes as C, time
import numpy as np
hip = C.CDLL("libamdhip64.so")


def ok(e):
    assert e == 0, e


N, SZ = 16, 781_000
ok(hip.hipSetDevice(0))
d = C.c_void_p(); ok(hip.hipMalloc(C.byref(d), C.c_size_t(N * (SZ + 4096))))
st = C.c_void_p(); ok(hip.hipStreamCreateWithFlags(C.byref(st), 1))
pinned = []
for i in range(N):
    p = C.c_void_p(); ok(hip.hipHostMalloc(C.byref(p), C.c_size_t(SZ), 0)); pinned.append(p.value)
bufs = [np.full(SZ, i, np.uint8) for i in range(N)]
for b in bufs:
    ok(hip.hipHostRegister(C.c_void_p(b.ctypes.data), C.c_size_t(SZ), 0))
one = C.c_void_p(); ok(hip.hipHostMalloc(C.byref(one), C.c_size_t(N * SZ), 0))
hip.hipMemcpyBatchAsync.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t,
                                    C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]


def timed(fn, reps=30):
    fn(); ok(hip.hipStreamSynchronize(st))
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); ok(hip.hipStreamSynchronize(st)); ts.append((time.perf_counter() - t0, t1 - t0))
    ts.sort()
    return ts[len(ts) // 2]


for name, srcs in (("pinned (hipHostMalloc)", pinned), ("registered ordinary memory", [b.ctypes.data for b in bufs])):
    dsts = [d.value + i * (SZ + 4096) for i in range(N)]

    def loop():
        for s, t in zip(srcs, dsts):
            ok(hip.hipMemcpyAsync(C.c_void_p(t), C.c_void_p(s), C.c_size_t(SZ), 1, st))
    A = (C.c_void_p * N)(*dsts); B = (C.c_void_p * N)(*srcs); S = (C.c_size_t * N)(*([SZ] * N)); fail = C.c_size_t(0)

    def batch():
        ok(hip.hipMemcpyBatchAsync(A, B, S, N, None, None, 0, C.byref(fail), st))
    for label, fn in (("loop of hipMemcpyAsync", loop), ("hipMemcpyBatchAsync", batch)):
        try:
            tot, call = timed(fn)
            print("%-28s %-24s %.3f ms (%.1f GB/s), host time in the calls %.3f ms" % (name, label, tot * 1e3, N * SZ / tot / 1e9, call * 1e3), flush=True)
        except AssertionError as e:
            print(name, label, "failed:", e, flush=True)
tot, call = timed(lambda: ok(hip.hipMemcpyAsync(d, one, C.c_size_t(N * SZ), 1, st)))
print("one copy of all %d x %d bytes: %.3f ms (%.1f GB/s)" % (N, SZ, tot * 1e3, N * SZ / tot / 1e9))
