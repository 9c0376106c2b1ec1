"""Steady-state rate of the transport-2 pipeline (set JGA_PIPE_TRACE=1 for per-group phase times).
Usage: e2e_trace.py [nimages] [batch] [lanes] [threads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jpeg_gpu_amd import abi, lib, synth
n, batch, lanes, threads = [int(a) for a in (sys.argv[1:] + ["960", "24", "4", "96"][len(sys.argv) - 1:])]
distinct = [synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1234 + i) for i in range(6)]
jobs = [distinct[i % 6] for i in range(n)]
pl = lib.Pipeline(device=0, nthreads=threads, out=abi.JPEG_DECODE_RGB, transport=2, batch=batch, depth=lanes)
pl.run(jobs)                                   # warm: the lanes' buffers, the clocks
t0 = time.perf_counter(); rc, _ = pl.run(jobs); dt = time.perf_counter() - t0
print("%d images, batch %d x %d lanes, %d threads: %.1f ms = %.1f Gpixel/s (rc %d)" % (n, batch, lanes, threads, dt * 1e3, n * 3840 * 2160 / dt / 1e9, rc))
pl.close()
