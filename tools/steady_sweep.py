#!/usr/bin/env python3
"""A >= 0.5 s stream of one geometry through the pipeline (the `steady` leg of tools/configs_bench.py) under any
jga_pipeline_config fields, pageable and pinned files: Mpixel/s, H2D GB/s, share of the link's ceiling.
    python tools/steady_sweep.py W H SAMPLING RI [cfg ...]      e.g.  tools/steady_sweep.py 1920 1080 420 0 "" "unstuff=2" """
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np                                   # noqa: E402
import oracle                                        # noqa: E402
from jpeg_gpu_amd import abi, lib, shard, synth      # noqa: E402
import configs_bench as cb                           # noqa: E402

w, h, samp, ri = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
cfgs = [eval("dict(%s)" % a) for a in sys.argv[5:]] or [{}]
files = [synth.synthetic_jpeg(w, h, samp, quality=90, seed=1234 + i, restart_interval=ri) for i in range(8)]
orc = oracle.Oracle()
quota = shard.cpu_quota()
cpus = len(os.sched_getaffinity(0))
budget = shard.rank_cpu_budget(cpus, 1, quota)
nthreads = min(cpus, max(8, budget + budget // 2)) if quota else max(1, min(cpus, 96))


class Lib:                                           # configs_bench's pipelines with extra fields
    def __init__(self, more):
        self.more = more

    def __getattr__(self, k):
        return getattr(lib, k)

    def Pipeline(self, **kw):
        kw.update(self.more)
        return lib.Pipeline(**kw)


Lib.Pipeline.make_jobs = lib.Pipeline.make_jobs
for cfg in cfgs:
    label = dict(cfg)
    group = cfg.pop("group", 32)
    lanes = cfg.pop("lanes", 8)
    seconds = cfg.pop("seconds", 0.6)
    keep = bool(cfg.pop("keep", 1))
    nfiles = cfg.pop("files", 8)
    for pinned in (False, True):
        r = cb._pipeline_steady(Lib(cfg), abi, np, orc, files[:nfiles], nthreads, group, lanes, seconds=seconds, pinned=pinned, keep=keep)
        print("%-40s %-8s %7.1f Gpixel/s  H2D %5.1f GB/s  %.3f of the link's ceiling  (%d images in %.3f s, ok %s)"
              % (label, "pinned" if pinned else "pageable", r["Mpixel_s"] / 1e3, r["h2d_GBps"], r["of_link_ceiling"], r["images"],
                 r["seconds"], r["bit_exact_vs_oracle"]), flush=True)
