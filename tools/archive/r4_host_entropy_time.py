"""Host entropy stage alone on the GPU box's host: ms per 4K 4:2:0 q90 frame, one thread and N threads
(the cgroup grants 16 CPUs), QUANT planes and PACK words."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from jpeg_gpu_amd import lib, synth
d = synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1234)
h, g = lib.geom_of(d)
for nthr in (1, 8, 16, 32, 48):
    outs = [np.zeros(g.coef_shorts, np.int16) for _ in range(nthr)]
    def work(i, reps=8):
        for _ in range(reps):
            assert lib.L.jga_entropy_decode(d, len(d), C.byref(g), outs[i].ctypes.data, 0) == 0
    with ThreadPoolExecutor(nthr) as ex:
        list(ex.map(lambda i: work(i, 1), range(nthr)))
        c0 = sum(os.times()[:2]); t0 = time.perf_counter()
        list(ex.map(work, range(nthr)))
        dt = time.perf_counter() - t0; cpu = sum(os.times()[:2]) - c0
    print("%2d threads: %.1f ms wall per frame per thread, %.1f ms of CPU per frame, %.1f CPUs busy, %.0f Mpixel/s" % (
        nthr, dt / 8 * 1e3, cpu / (8 * nthr) * 1e3, cpu / dt, nthr * 8 * 3840 * 2160 / dt / 1e6), flush=True)
