import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if "--torch" in sys.argv:
    import torch
    torch.cuda.init()
from jpeg_gpu_amd import abi, lib, synth
files = [synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=s) for s in range(16)]
pins = [lib.PinnedBytes(f) for f in files]
import numpy as np
def measure(pinned, reps=15, fresh_copy=False, **cfg):
    pl = lib.Pipeline(device=0, nthreads=24, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2, batch=32, depth=8, **cfg)
    src = [p.array for p in pins] if pinned else ([np.frombuffer(f, np.uint8).copy() for f in files] if fresh_copy else files)
    jobs = lib.Pipeline.make_jobs([src[i % 16] for i in range(128)], pinned=pinned)
    pl.run_jobs(jobs)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); rc = pl.run_jobs(jobs); ts.append((time.perf_counter() - t0) * 1e3); assert rc == 0
    c = pl.counters()
    pl.close()
    ts.sort()
    return "min %.2f med %.2f  (registered %d, in place %d, copied %d)" % (ts[0], ts[len(ts) // 2], c["registered"], c["jobs_in_place"], c["jobs_copied"])
for label, kw in (("pageable (bytes objects)", dict(pinned=False)), ("pageable (numpy copies)", dict(pinned=False, fresh_copy=True)),
                  ("pinned", dict(pinned=True)), ("pageable, no cache (-1)", dict(pinned=False, input_cache_mb=-1)),
                  ("pageable, host copies (-2)", dict(pinned=False, input_cache_mb=-2)), ("pageable again", dict(pinned=False))):
    print("%-28s %s" % (label, measure(**kw)), flush=True)
