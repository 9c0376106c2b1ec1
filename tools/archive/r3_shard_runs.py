"""rank 3's 128-file shard of config 4 through the pipeline, every run's time (not the best of n)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jpeg_gpu_amd import abi, lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
files = [synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=s) for s in range(16)]
pl = lib.Pipeline(device=0, nthreads=24, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2, batch=32, depth=8)
jobs = lib.Pipeline.make_jobs([files[i % 16] for i in range(n)])
ts = []
for _ in range(12):
    t0 = time.perf_counter(); pl.run_jobs(jobs); ts.append((time.perf_counter() - t0) * 1e3)
print("no trace:", " ".join("%.2f" % t for t in ts))
os.environ["JGA_PIPE_TRACE"] = "1"
ts = []
for _ in range(6):
    t0 = time.perf_counter(); pl.run_jobs(jobs); ts.append((time.perf_counter() - t0) * 1e3)
print("trace:", " ".join("%.2f" % t for t in ts))
del os.environ["JGA_PIPE_TRACE"]
ts = []
for _ in range(6):
    t0 = time.perf_counter(); pl.run_jobs(jobs); ts.append((time.perf_counter() - t0) * 1e3)
print("no trace again:", " ".join("%.2f" % t for t in ts))
