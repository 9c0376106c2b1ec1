"""Would decoding a batch as two halves on two streams beat one batch?  Two HuffBatch objects of n/2 images
decoded from two threads at once, against one of n."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jpeg_gpu_amd import lib, synth
n = 48
files = [synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1234 + i) for i in range(6)]
jp = [files[i % 6] for i in range(n)]


def mk(jobs):
    hb = lib.HuffBatch(len(jobs), sum(map(len, jobs)) + 4096 * len(jobs))
    g = hb.prepare(jobs)
    cs = (g.coef_shorts * 2 + 255) // 256 * 128
    dcs = (g.coef_shorts // 64 + 127) // 128 * 128
    return hb, lib.DeviceBuffer(cs * 2 * len(jobs)), cs, lib.DeviceBuffer(dcs * 2 * len(jobs)), dcs, lib.L.jga_stream_create()


one = mk(jp)
ha, hb2 = mk(jp[:n // 2]), mk(jp[n // 2:])
lib.check(lib.L.jga_stream_sync(None))


def dec(x):
    x[0].decode_split(x[1].ptr, x[2], x[3].ptr, x[4], x[5])


for rep in range(4):
    t0 = time.perf_counter(); dec(one); t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    ts = [threading.Thread(target=dec, args=(h,)) for h in (ha, hb2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    t2 = time.perf_counter() - t0
    t0 = time.perf_counter(); dec(ha); dec(hb2); t3 = time.perf_counter() - t0
    print("one batch of %d: %.3f ms | two halves at once: %.3f ms | two halves one after the other: %.3f ms" % (n, t1 * 1e3, t2 * 1e3, t3 * 1e3))
