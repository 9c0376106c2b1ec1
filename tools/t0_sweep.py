"""Host-Huffman transports (0: dense planes, 1: PACK words) over the number of worker threads."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from jpeg_gpu_amd import abi, lib, synth
W, H = 3840, 2160
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(16) as ex:
    jpegs = list(ex.map(lambda s: synth.synthetic_jpeg(W, H, "420", 90, seed=s), range(64)))
lib.check(lib.L.jga_set_device(0))
for transport in (0, 1):
    for nthr in [int(a) for a in (sys.argv[1:] or "16 32 48 64 96 128 192".split())]:
        pl = lib.Pipeline(device=0, nthreads=nthr, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=transport)
        n = max(192, 6 * nthr)
        cyc = lambda k, o=0: [jpegs[(o + i) % 64] for i in range(k)]
        pl.run_jobs(lib.Pipeline.make_jobs(cyc(2 * nthr)))
        jobs = lib.Pipeline.make_jobs(cyc(n, 5))
        t0 = time.perf_counter(); rc = pl.run_jobs(jobs); dt = time.perf_counter() - t0
        pl.close()
        print("transport %d threads %3d: %8.1f Mpix/s (%d images, rc %d)" % (transport, nthr, n * W * H / dt / 1e6, n, rc), flush=True)
