#!/usr/bin/env python3
"""One traced run of config 4's 128-file shard (jga_pipeline_config.trace = 1): per-group phase times on stderr.
    python tools/shard_trace.py [n] [cfg=value ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jpeg_gpu_amd import abi, lib, synth          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = eval("dict(%s)" % ",".join(sys.argv[2:])) if len(sys.argv) > 2 else {}
pinned = cfg.pop("pinned", 0)
files = [synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=s) for s in range(16)]
pins = [lib.PinnedBytes(f) for f in files]
src = [p.array for p in pins] if pinned else files
jobs = lib.Pipeline.make_jobs([src[i % 16] for i in range(n)], pinned=bool(pinned))
pl = lib.Pipeline(device=0, nthreads=24, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2, batch=32, depth=8, **cfg)
for _ in range(10):
    pl.run_jobs(jobs)
pl.close()
pl = lib.Pipeline(device=0, nthreads=24, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2, batch=32, depth=8, trace=1, **cfg)
for _ in range(6):
    pl.run_jobs(jobs)
print("==== traced runs", cfg, "pinned" if pinned else "pageable", file=sys.stderr, flush=True)
for _ in range(3):
    t0 = time.perf_counter()
    pl.run_jobs(jobs)
    print("TOTAL %.2f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
pl.close()
