"""Does a long-lived process see the pageable shard differently, and is it the host copies of the waiting groups?  (tuning build)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from jpeg_gpu_amd import abi, lib, synth
import numpy as np
files = [synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=s) for s in range(16)]
pins = [lib.PinnedBytes(f) for f in files]
def measure(pinned, reps=15, **cfg):
    pl = lib.Pipeline(device=0, nthreads=24, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2, batch=32, depth=8, **cfg)
    src = [p.array for p in pins] if pinned else files
    jobs = lib.Pipeline.make_jobs([src[i % 16] for i in range(128)], pinned=pinned)
    for _ in range(8):
        pl.run_jobs(jobs)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); rc = pl.run_jobs(jobs); ts.append((time.perf_counter() - t0) * 1e3); assert rc == 0
    pl.close()
    ts.sort()
    return "min %.2f med %.2f p80 %.2f" % (ts[0], ts[len(ts) // 2], ts[int(len(ts) * 0.8)])
def age(seconds):
    pl = lib.Pipeline(device=0, nthreads=24, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2, batch=32, depth=8)
    jobs = lib.Pipeline.make_jobs([files[i % 16] for i in range(1024)])
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        pl.run_jobs(jobs)
    pl.close()
for phase in ("fresh process", "after 6 s of work", "after 20 s of work"):
    if phase != "fresh process":
        age(6 if "6" in phase else 14)
    for cw in ("1", "0"):
        os.environ["JGA_PIPE_COPY_WAITING"] = cw
        print("%-20s copy-while-waiting %s: pageable %s | pinned %s" % (phase, cw, measure(False), measure(True)), flush=True)
try:
    print("numa_balancing:", open("/proc/sys/kernel/numa_balancing").read().strip())
except Exception as e:
    print("numa_balancing: ?", e)
