import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from concurrent.futures import ThreadPoolExecutor
from jpeg_gpu_amd import abi, lib, synth
W, H = 3840, 2160
mode = int(os.environ.get("UNSTUFF", "2")); keep = os.environ.get("KEEP", "1") == "1"; setup_n = int(os.environ.get("SETUP", "256"))
with ThreadPoolExecutor(16) as ex:
    files = list(ex.map(lambda s: synth.synthetic_jpeg(W, H, "420", 90, seed=1234 + s), range(64)))
pins = [lib.PinnedBytes(d) for d in files]
_, g = lib.geom_of(files[0])
n = 2560
pitch = (g.rgb_bytes + 255) // 256 * 256
buf = torch.empty(n * pitch, dtype=torch.uint8, device="cuda") if keep else None
pl = lib.Pipeline(device=0, nthreads=24, out=abi.JPEG_DECODE_RGB, transport=2, batch=32, depth=8, unstuff=mode)
pc = lambda k, o=0: [pins[(o + i) % 64].array for i in range(k)]
setup = lib.Pipeline.make_jobs(pc(setup_n), pinned=True)
pl.run_jobs(setup)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    pl.run_jobs(setup)
pl.run_jobs(lib.Pipeline.make_jobs(pc(640, 7), pinned=True))
jobs = lib.Pipeline.make_jobs(pc(n, 13), pinned=True, dev_outs=[buf.data_ptr() + k * pitch for k in range(n)] if keep else None)
torch.cuda.synchronize()
t0 = time.perf_counter(); rc = pl.run_jobs(jobs); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("UNSTUFF=%d KEEP=%d SETUP=%d: %.1f ms = %.1f Gpix/s rc %d" % (mode, keep, setup_n, dt * 1e3, n * W * H / dt / 1e9, rc))
