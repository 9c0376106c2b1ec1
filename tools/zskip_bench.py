"""Fused IDCT+RGB kernel on dense and on sparse (photograph-like) coefficient planes: time per
48 x 4K launch, and equality with the oracle.  Usage: python tools/zskip_bench.py [reps]
(library variant via JGA_LIB_PATH; A/B of the exact zero-skipping transform)."""
import ctypes as C, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image, ImageFile
ImageFile.MAXBLOCK = 1 << 26
import oracle
from jpeg_gpu_amd import lib, synth
W, H, N = 3840, 2160, 48
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(0)
x = np.linspace(0, 1, W)[None, :, None]; y = np.linspace(0, 1, H)[:, None, None]; c = np.arange(3)[None, None, :]


def pil(img, q, sub):
    buf = io.BytesIO()
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=q, subsampling=sub)
    return buf.getvalue()


discs = np.zeros((H, W, 3), np.float32)
for _ in range(60):
    cx, cy, r = rng.random(), rng.random(), rng.random()*0.2
    discs += (((x - cx)**2 + (y - cy)**2) < r*r)*rng.integers(-80, 80, 3)
cases = {
    "bench synth q90 (dense)": synth.synthetic_jpeg(W, H, "420", 90, seed=1234),
    "smooth + grain(2) q90": pil(128 + 60*np.sin(6*x*(c + 1))*np.cos(4*y) + rng.normal(0, 2, (H, W, 3)), 90, 2),
    "discs + grain(4) q85": pil(128 + discs + rng.normal(0, 4, (H, W, 3)), 85, 2),
    "discs + grain(4) q85 4:4:4": pil(128 + discs + rng.normal(0, 4, (H, W, 3)), 85, 0),
    "flat grey": pil(np.full((H, W, 3), 120.0), 90, 2),
}
orc = oracle.Oracle()
lib.check(lib.L.jga_set_device(0))
for name, data in cases.items():
    hd, g = lib.geom_of(data)
    n = N if g.subsamp != 1 else 24
    coef = lib.entropy_decode(data, g)
    nz = np.count_nonzero(coef)/g.coef_blocks
    cs = (g.coef_shorts*2 + 255)//256*128; os_ = (g.rgb_bytes + 255)//256*256
    dc, dq, do = lib.DeviceBuffer(cs*2*n), lib.DeviceBuffer(384*n), lib.DeviceBuffer(os_*n)
    for i in range(n):
        dc.upload(coef, offset=i*cs*2)
    dq.upload(np.tile(lib.qtab_of(hd).reshape(-1), n))
    ms = C.c_float()
    for r in (10, reps):
        lib.check(lib.L.jga_time_idct_batch(C.byref(g), n, dc.ptr, cs, dq.ptr, 1, do.ptr, os_, 1, r, None, C.byref(ms)))
    same = bool(np.array_equal(do.download(g.rgb_bytes, offset=(n - 1)*os_), orc.decode_rgb(data)[1].reshape(-1)))
    alg = n*(g.coef_blocks*128 + g.rgb_bytes)
    print("%-28s %4.1f nonzero/block  x%d: %.4f ms  %5.0f GB/s  %6.0f Gpixel/s  equals oracle: %s" % (
        name, nz, n, ms.value, alg/ms.value/1e6, n*W*H/ms.value/1e6, same), flush=True)
    dc.free(); dq.free(); do.free()
