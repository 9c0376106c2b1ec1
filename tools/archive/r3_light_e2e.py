"""Pipeline (transport 2) on lighter content and on the bench recipe: Gpixel/s, best of 3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from jpeg_gpu_amd import abi, lib, synth
W, H = 3840, 2160


def photo_like(i):
    r = np.random.default_rng(900 + i)
    xx = np.linspace(0, 1, W, dtype=np.float32)[None, :, None]
    yy = np.linspace(0, 1, H, dtype=np.float32)[:, None, None]
    cc = np.arange(3, dtype=np.float32)[None, None, :]
    img = 128 + 60 * np.sin((6 + i) * xx * (cc + 1)) * np.cos(4 * yy) + r.normal(0, 2, (H, W, 3)).astype(np.float32)
    return synth.encode_pixels(np.clip(img, 0, 255).astype(np.uint8), "420", 90)


with ThreadPoolExecutor(8) as ex:
    light = list(ex.map(photo_like, range(8)))
    heavy = list(ex.map(lambda s: synth.synthetic_jpeg(W, H, "420", 90, seed=1234 + s), range(16)))
for name, files, n in (("light", light, 1920), ("bench", heavy, 1536)):
    pl = lib.Pipeline(device=0, nthreads=24, out=abi.JPEG_DECODE_RGB, transport=2, batch=int(os.environ.get("BATCH", "32")), depth=int(os.environ.get("LANES", "8")))
    jobs = lib.Pipeline.make_jobs([files[i % len(files)] for i in range(n)])
    pl.run_jobs(jobs)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); rc = pl.run_jobs(jobs); best = min(best, time.perf_counter() - t0)
    pl.close()
    print("%s %.1f" % (name, n * W * H / best / 1e9), end="  ")
print()
