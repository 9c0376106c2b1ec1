import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jpeg_gpu_amd import abi, lib, shard, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pinned = len(sys.argv) > 2 and sys.argv[2] == "pin"
files = [synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=s) for s in range(16)]
pins = [lib.PinnedBytes(f) for f in files]
src = [p.array for p in pins] if pinned else files
pl = lib.Pipeline(device=0, nthreads=24, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2, batch=32, depth=8)
jobs = lib.Pipeline.make_jobs([src[i % 16] for i in range(n)], pinned=pinned)
for _ in range(3):
    pl.run_jobs(jobs)
os.environ["JGA_PIPE_TRACE"] = "1"
for _ in range(3):
    t0 = time.perf_counter(); pl.run_jobs(jobs); print("TOTAL %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
