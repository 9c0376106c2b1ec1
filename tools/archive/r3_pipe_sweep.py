"""Pipeline (transport 2) on short and long jobs for one setting of the environment knobs:
1024 x 1080p, rank 3's 128-file shard, 96 x 4K 4:4:4, 1536 x 4K 4:2:0.  Prints Gpixel/s (best of 3)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jpeg_gpu_amd import abi, lib, shard, synth  # noqa: E402

quota = shard.cpu_quota()
cpus = len(os.sched_getaffinity(0))
budget = shard.rank_cpu_budget(cpus, 1, quota)
nthreads = min(cpus, max(8, budget + budget // 2)) if quota else max(1, min(cpus, 96))


def run(files, order, pinned=False, reps=3):
    _, g = lib.geom_of(files[0])
    n = len(order)
    ostride = (g.rgb_bytes + 255) // 256 * 256
    out = lib.DeviceBuffer(ostride * min(n, 256))
    pins = [lib.PinnedBytes(f) for f in files] if pinned else None
    src = [p.array for p in pins] if pinned else files
    pl = lib.Pipeline(device=0, nthreads=nthreads, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2, batch=32, depth=8)
    jobs = lib.Pipeline.make_jobs([src[i] for i in order], pinned=pinned)
    pl.run_jobs(jobs)
    best = 1e9
    for _ in range(reps):
        lib.check(lib.L.jga_stream_sync(None))
        t0 = time.perf_counter()
        rc = pl.run_jobs(jobs)
        best = min(best, time.perf_counter() - t0)
        assert rc == 0
    pl.close()
    out.free()
    if pins:
        for p in pins:
            p.free()
    return n * g.width * g.height / best / 1e9, best * 1e3


f1080 = [synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=s) for s in range(16)]
f444 = [synth.synthetic_jpeg(3840, 2160, "444", quality=90, seed=1234 + s) for s in range(4)]
f4k = [synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1234 + s) for s in range(16)]
res = []
for name, files, order in (("1024x1080p", f1080, [i % 16 for i in range(1024)]),
                           ("shard128", f1080, [i % 16 for i in shard.shard_range(1024, 3, 8)]),
                           ("96x4K444", f444, [i % 4 for i in range(96)]),
                           ("1536x4K420", f4k, [i % 16 for i in range(1536)])):
    for pinned in (False, True):
        gp, ms = run(files, order, pinned)
        res.append("%s%s %.1f (%.2f ms)" % (name, "/pin" if pinned else "", gp, ms))
print(" | ".join(res))
