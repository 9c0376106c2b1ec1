"""North-star transports (0: dense planes, 1: PACK words) after the host entropy stage's rewrite: Mpixel/s and CPUs
in use (process CPU seconds / wall) by host thread count and job length.  Usage: r4_ns_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
from jpeg_gpu_amd import abi, lib, synth
files = [synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1234 + i) for i in range(16)]
px = 3840 * 2160
for transport in (0, 1):
    for nthr in (16, 24, 32, 48, 64):
        for n in (192, 768):
            pl = lib.Pipeline(device=0, nthreads=nthr, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=transport)
            jobs = lib.Pipeline.make_jobs([files[i % 16] for i in range(n)])
            pl.run_jobs(lib.Pipeline.make_jobs([files[i % 16] for i in range(2 * nthr)]))
            best = None
            for _ in range(2):
                c0 = sum(os.times()[:2]); t0 = time.perf_counter()
                rc = pl.run_jobs(jobs)
                dt = time.perf_counter() - t0; cpu = sum(os.times()[:2]) - c0
                assert rc == 0
                if best is None or dt < best[0]:
                    best = (dt, cpu)
            pl.close()
            print("transport %d threads %2d images %3d: %6.0f Mpixel/s, %.1f CPUs busy, %.1f ms of CPU per image, H2D %.1f GB/s" % (
                transport, nthr, n, n * px / best[0] / 1e6, best[1] / best[0], best[1] / n * 1e3,
                sum(j.h2d_bytes for j in jobs) / best[0] / 1e9), flush=True)
