"""Does config 4's 128-file shard get slower with what the process has done before (bench.py's context: 4.4 ms
pageable where a fresh process measures 3.4)?  The shard leg of tools/configs_bench.py, repeated between the other legs
it runs.  Usage: python tools/archive/r5_shard_aging.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle
from jpeg_gpu_amd import abi, lib, shard, synth
import configs_bench as cb
orc = oracle.Oracle()
files = [synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=s) for s in range(16)]
order128 = [i % 16 for i in shard.shard_range(1024, 3, 8)]


def shard_leg(tag):
    for kind, pinned in (("pageable", False), ("pinned", True)):
        ts = []
        dt, ok, h2d = cb._pipeline_stream(lib, abi, np, orc, files, order128, 24, 32, 8, reps=15, pinned=pinned, times=ts)
        ts.sort()
        print("%-46s %-8s min %.2f med %.2f max %.2f ok %s" % (tag, kind, ts[0] * 1e3, ts[7] * 1e3, ts[-1] * 1e3, ok), flush=True)


shard_leg("fresh process")
shard_leg("again")
ts = []
cb._pipeline_stream(lib, abi, np, orc, files, [i % 16 for i in range(1024)], 24, 32, 8, reps=3, pinned=False, times=ts)
print("1024 files pageable: %s" % " ".join("%.2f" % (t * 1e3) for t in ts), flush=True)
shard_leg("after the 1024-file job (pageable)")
cb._pipeline_stream(lib, abi, np, orc, files, [i % 16 for i in range(1024)], 24, 32, 8, reps=3, pinned=True, times=ts)
shard_leg("after the 1024-file job (pinned)")
f4k = [synth.synthetic_jpeg(3840, 2160, "444", quality=90, seed=1234 + i) for i in range(4)]
cb._pipeline_stream(lib, abi, np, orc, f4k, [i % 4 for i in range(96)], 24, 32, 8, reps=2)
shard_leg("after 96 x 4K 4:4:4")
st = cb._pipeline_steady(lib, abi, np, orc, files, 24, 32, 8) if hasattr(cb, "_pipeline_steady") else None
print("steady leg:", st, flush=True)
shard_leg("after a steady leg of >= 0.5 s")
print(cb._plugin(lib, abi, files[0], 10), flush=True)
shard_leg("after the plugin leg")
print(cb._device_only(lib, files, 1024, 3), flush=True)
shard_leg("after device-only 1024")
for pause in (0.0, 0.3, 0.0, 1.0):
    cb._pipeline_stream(lib, abi, np, orc, files, [i % 16 for i in range(1024)], 24, 32, 8, reps=3, pinned=False, times=ts)
    time.sleep(pause)
    shard_leg("after the 1024-file job + %.1f s of sleep" % pause)
