"""JPEG bytes in host RAM -> RGB in HBM through jga_pipeline transport 2, swept over
(batch, lanes, host threads).  Usage: python tools/e2e_sweep.py [nimages]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jpeg_gpu_amd import abi, lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
W, H = 3840, 2160
distinct = [synth.synthetic_jpeg(W, H, "420", quality=90, seed=1234 + i) for i in range(6)]
jobs = [distinct[i % 6] for i in range(n)]
for batch, depth, nthr in [(16, 3, 48), (16, 4, 64), (24, 3, 48), (24, 4, 96), (32, 3, 96), (48, 2, 96),
                           (12, 6, 96), (8, 8, 64), (16, 6, 96)]:
    pl = lib.Pipeline(device=0, nthreads=nthr, out=abi.JPEG_DECODE_RGB, transport=2, batch=batch,
                      depth=depth)
    pl.run(jobs[:batch * depth])
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        rc, _ = pl.run(jobs)
        best = min(best, time.perf_counter() - t0)
    pl.close()
    print("batch %2d lanes %d threads %2d: %.1f ms for %d images = %.1f Gpixel/s (rc %d)" % (
        batch, depth, nthr, best * 1e3, n, n * W * H / best / 1e9, rc), flush=True)
