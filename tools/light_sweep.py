#!/usr/bin/env python3
"""bench.py's lighter-content leg (smooth 4K image with fine grain, 0.6 bits per pixel) through the pipeline under
any jga_pipeline_config fields: Gpixel/s of 3 runs of N frames each.   python tools/light_sweep.py [N] [cfg ...]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                   # noqa: E402
from jpeg_gpu_amd import abi, lib, synth          # noqa: E402

W, H = 3840, 2160
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2560


def photo_like(i):
    r = np.random.default_rng(900 + i)
    xx = np.linspace(0, 1, W, dtype=np.float32)[None, :, None]
    yy = np.linspace(0, 1, H, dtype=np.float32)[:, None, None]
    cc = np.arange(3, dtype=np.float32)[None, None, :]
    img = 128 + 60 * np.sin((6 + i) * xx * (cc + 1)) * np.cos(4 * yy) + r.normal(0, 2, (H, W, 3)).astype(np.float32)
    return synth.encode_pixels(np.clip(img, 0, 255).astype(np.uint8), "420", 90)


with ThreadPoolExecutor(max_workers=8) as ex:
    light = list(ex.map(photo_like, range(8)))
variants = [eval("dict(%s)" % a) for a in sys.argv[2:]] or [{}]
for v in variants:
    kw = dict(dict(device=0, nthreads=24, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2, batch=32, depth=8), **v)
    pl = lib.Pipeline(**kw)
    jobs = lib.Pipeline.make_jobs([light[i % 8] for i in range(n)])
    pl.run_jobs(jobs)
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        assert pl.run_jobs(jobs) == 0
        rates.append(n * W * H / (time.perf_counter() - t0) / 1e9)
    pl.close()
    print("%-40s %s Gpixel/s" % (v, " ".join("%.0f" % r for r in rates)), flush=True)
