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
for transport in (2, 0, 1):
    base = None
    for r in range(reps):
        pl = lib.Pipeline(device=0, nthreads=16, out=abi.JPEG_DECODE_RGB, transport=transport, batch=8, depth=4)
        rc, _ = pl.run(jobs)
        assert rc == 0
        pl.close()
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        used = (total - free) / 2**20
        if r == 1:
            base = (used, rss_mb())
        if r in (1, reps - 1):
            print("transport %d, rep %2d: device used %.0f MB, RSS %.0f MB" % (transport, r, used, rss_mb()))
    used, rss = (total - free) / 2**20, rss_mb()
    assert used - base[0] < 64, "device memory grows"
    assert rss - base[1] < 256, "host memory grows"
print("no growth")
