"""Create / run / destroy the pipeline and the GPU entropy batch repeatedly: device and host memory
must come back (hipMemGetInfo through torch, RSS from /proc).  Usage: leak_check.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (first: its HIP runtime is the one both bind to)
from jpeg_gpu_amd import abi, lib, synth  # noqa: E402


def rss_mb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) / 1024


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
jpegs = [synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=i) for i in range(8)]
jobs = [jpegs[i % 8] for i in range(192)]
import numpy as np  # noqa: E402
arrs = [np.frombuffer(j, np.uint8).copy() for j in jpegs]
# (round 4: also the input cache — registrations must be dropped by destroy — and the upload in pieces,
# whose batch objects own copy / kernel streams and events of their own)
configs = ((2, {}), (2, dict(unstuff=2, input_cache_mb=64)), (2, dict(batch=48, unstuff=2, input_cache_mb=-1)),
           (2, dict(batch=48, unstuff=1)), (0, {}), (1, {}))
if len(sys.argv) > 2:                      # e.g. "1" or "1,0": only these transports, plain
    configs = tuple((int(t), {}) for t in sys.argv[2].split(","))
for transport, more in configs:
    base = None
    for r in range(reps):
        pl = lib.Pipeline(**dict(dict(device=0, nthreads=16, out=abi.JPEG_DECODE_RGB, transport=transport, batch=8, depth=4), **more))
        rc = pl.run_jobs(lib.Pipeline.make_jobs([arrs[i % 8] for i in range(192)])) if more else pl.run(jobs)[0]
        assert rc == 0
        pl.close()
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        used = (total - free) / 2**20
        if r == (5 if reps >= 10 else 1):     # (the runtime's pinned-memory pools take a step up within the first few
            base = (used, rss_mb())            #  creations and stay there: 41 repetitions of one transport, round 4)
        if r in (1, reps - 1) or (len(sys.argv) > 2 and r % 5 == 0):
            print("transport %d %s, rep %2d: device used %.0f MB, RSS %.0f MB" % (transport, more, r, used, rss_mb()))
    used, rss = (total - free) / 2**20, rss_mb()
    assert used - base[0] < 64, "device memory grows"
    assert rss - base[1] < 256, "host memory grows"
print("no growth")
