"""Short jobs through the pipeline for one setting of the environment knobs, every run's time:
rank 3's 128-file shard and all 1024 files of config 4 (engines warmed first)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jpeg_gpu_amd import abi, lib, synth
files = [synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=s) for s in range(16)]
pl = lib.Pipeline(device=0, nthreads=24, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2, batch=32, depth=8)
res = []
for n in (128, 1024):
    jobs = lib.Pipeline.make_jobs([files[i % 16] for i in range(n)])
    for _ in range(10 if n == 128 else 3):
        pl.run_jobs(jobs)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); rc = pl.run_jobs(jobs); ts.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0
    ts.sort()
    res.append("%d files: median %.2f ms = %.1f Gpixel/s (best %.2f)" % (n, ts[4], n * 1920 * 1080 / ts[4] / 1e6, ts[0]))
print(" | ".join(res))
