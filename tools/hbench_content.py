"""GPU entropy stage on other content than the bench recipe: photograph-like 4K files made with
Pillow (smooth + grain, discs + grain) next to the bench images.  48 files resident in HBM ->
planes; time per batch and symbols.  Usage: python tools/hbench_content.py"""
import io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image, ImageFile
ImageFile.MAXBLOCK = 1 << 26
from jpeg_gpu_amd import lib, synth
W, H, N = 3840, 2160, 48
rng = np.random.default_rng(0)
x = np.linspace(0, 1, W)[None, :, None]; y = np.linspace(0, 1, H)[:, None, None]; c = np.arange(3)[None, None, :]
def pil(img, q):
    buf = io.BytesIO(); Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=q, subsampling=2); return buf.getvalue()
discs = np.zeros((H, W, 3), np.float32)
for _ in range(60):
    cx, cy, r = rng.random(), rng.random(), rng.random()*0.2
    discs += (((x - cx)**2 + (y - cy)**2) < r*r)*rng.integers(-80, 80, 3)
cases = {"bench synth q90": [synth.synthetic_jpeg(W, H, "420", 90, seed=1234 + i) for i in range(4)],
         "smooth + grain(2) q90": [pil(128 + 60*np.sin(6*x*(c + 1))*np.cos(4*y) + rng.normal(0, 2, (H, W, 3)), 90) for i in range(2)],
         "discs + grain(4) q85": [pil(128 + discs + rng.normal(0, 4, (H, W, 3)), 85) for i in range(2)],
         "discs + grain(8) q92": [pil(128 + discs + rng.normal(0, 8, (H, W, 3)), 92) for i in range(2)]}
lib.check(lib.L.jga_set_device(0))
for name, files in cases.items():
    jobs = [files[i % len(files)] for i in range(N)]
    hb = lib.HuffBatch(N, sum(map(len, jobs)) + 4096*N)
    g = hb.prepare(jobs)
    stride = (g.coef_shorts*2 + 255)//256*128
    d = lib.DeviceBuffer(stride*2*N)
    ts = []
    for rep in range(4):
        t0 = time.perf_counter(); rounds = hb.decode(d.ptr, stride); ts.append(time.perf_counter() - t0)
    got = d.download(g.coef_shorts*2, dtype=np.int16); want = lib.entropy_decode(jobs[0], g); m = lib.real_coef_mask(g)
    print("%-24s %5.2f MB/file: huffman %.3f ms (%d rounds) = %6.0f Mpixel/s  equal host stage: %s" % (
        name, len(jobs[0])/1e6, min(ts[1:])*1e3, rounds, N*W*H/min(ts[1:])/1e6, bool(np.array_equal(got[m], want[m]))), flush=True)
    hb.close(); d.free()
