#!/usr/bin/env python3
"""Config 4's 128-file shard (what one rank of 8 gets) through the pipeline under different scheduling
fields of jga_pipeline_config: min / median of 15 runs each, pageable and pinned files.
    python tools/shard_sweep.py [n_files]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jpeg_gpu_amd import abi, lib, synth          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
files = [synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=s) for s in range(16)]
pins = [lib.PinnedBytes(f) for f in files]
px = n * 1920 * 1080


def measure(pinned, reps=15, **cfg):
    pl = lib.Pipeline(device=0, nthreads=24, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2, batch=32, depth=8, **cfg)
    src = [p.array for p in pins] if pinned else files
    jobs = lib.Pipeline.make_jobs([src[i % 16] for i in range(n)], pinned=pinned)
    for _ in range(8):
        pl.run_jobs(jobs)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        rc = pl.run_jobs(jobs)
        ts.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0
    pl.close()
    ts.sort()
    return ts[0], ts[len(ts) // 2]


variants = [
    {}, {"spin_waits": 1}, {"unstuff": 1}, {"unstuff": 2}, {"input_cache_mb": -1}, {"input_cache_mb": -2},
]
extra = [eval("dict(%s)" % a) for a in sys.argv[2:]]
for v in (extra or variants):
    a, b = measure(False, **v), measure(True, **v)
    print("%-44s pageable min %.2f med %.2f ms (%.1f Gpx/s) | pinned min %.2f med %.2f ms (%.1f Gpx/s)"
          % (v, a[0], a[1], px / a[1] / 1e6, b[0], b[1], px / b[1] / 1e6), flush=True)
