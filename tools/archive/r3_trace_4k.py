import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jpeg_gpu_amd import abi, lib, shard, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
files = [synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1234 + s) for s in range(16)]
pl = lib.Pipeline(device=0, nthreads=24, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2, batch=32, depth=8)
jobs = lib.Pipeline.make_jobs([files[i % 16] for i in range(n)])
for _ in range(2):
    t0 = time.perf_counter(); pl.run_jobs(jobs); dt = time.perf_counter() - t0
    print("run %.1f ms = %.1f Gpix/s" % (dt * 1e3, n * 3840 * 2160 / dt / 1e9))
os.environ["JGA_PIPE_TRACE"] = "1"
jobs2 = lib.Pipeline.make_jobs([files[i % 16] for i in range(256)])
t0 = time.perf_counter(); pl.run_jobs(jobs2); print("TOTAL %.2f ms" % ((time.perf_counter() - t0) * 1e3))
