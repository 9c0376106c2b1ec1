"""GPU entropy stage benchmark: JPEG bytes resident in HBM -> coefficient planes -> RGB.
Usage: python tools/hbench.py [W H sampling nimages restart_interval]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from jpeg_gpu_amd import lib, synth  # noqa: E402

w, h, samp, n, ri = (sys.argv[1:] + [None] * 5)[:5]
w, h, samp, n, ri = int(w or 3840), int(h or 2160), samp or "420", int(n or 48), int(ri or 0)
if os.environ.get("CONTENT") == "light":       # bench.py's lighter content: smooth image + fine grain, 0.6 bits per pixel
    def photo_like(i):
        r = np.random.default_rng(900 + i)
        xx = np.linspace(0, 1, w, dtype=np.float32)[None, :, None]
        yy = np.linspace(0, 1, h, dtype=np.float32)[:, None, None]
        cc = np.arange(3, dtype=np.float32)[None, None, :]
        img = 128 + 60 * np.sin((6 + i) * xx * (cc + 1)) * np.cos(4 * yy) + r.normal(0, 2, (h, w, 3)).astype(np.float32)
        return synth.encode_pixels(np.clip(img, 0, 255).astype(np.uint8), samp, 90, restart_interval=ri)
    distinct = [photo_like(i) for i in range(min(n, 6))]
elif os.environ.get("CONTENT") == "photo":     # round 6's tracked content class: f^-1.5 spectrum + grain, 0.14-0.17 B/px
    distinct = [synth.photo_like_jpeg(w, h, samp, 90, ri, seed=1 + i) for i in range(min(n, 4))]
else:
    distinct = [synth.synthetic_jpeg(w, h, samp, quality=90, restart_interval=ri, seed=1234 + i)
                for i in range(min(n, 6))]
jpegs = [distinct[i % len(distinct)] for i in range(n)]
hb = lib.HuffBatch(n, sum(map(len, jpegs)) + 4096 * n)
if os.environ.get("PINNED") == "1":        # files in pinned memory, scans DMA'd in place + cleaned up on the GPU
    pins = [lib.PinnedBytes(j) for j in distinct]
    lib.L.jga_huff_set_device_unstuff(hb.ptr, 1)
    lib.L.jga_huff_set_inputs_pinned(hb.ptr, 1)
if os.environ.get("THREADS"):
    lib.L.jga_huff_set_threads(hb.ptr, int(os.environ["THREADS"]))
preps = []
for _ in range(4):
    t0 = time.perf_counter()
    if os.environ.get("PINNED") == "1":
        g = hb.prepare_at([pins[i % len(pins)].array.ctypes.data for i in range(n)], [len(j) for j in jpegs])
    else:
        g = hb.prepare(jpegs)
    t1 = time.perf_counter()
    lib.check(lib.L.jga_stream_sync(None))
    preps.append((t1 - t0, time.perf_counter() - t0))
t_prep = preps[-1][1]
print("prepare host / host+H2D ms:", ["%.1f/%.1f" % (a * 1e3, b * 1e3) for a, b in preps])
stride = (g.coef_shorts * 2 + 255) // 256 * 128
ostride = (g.rgb_bytes + 255) // 256 * 256
d_coef = lib.DeviceBuffer(stride * 2 * n)
d_q = lib.DeviceBuffer(3 * 64 * 2 * n)
d_rgb = lib.DeviceBuffer(ostride * n)
d_q.upload(hb.qtabs())
split = os.environ.get("SPLIT", "1") == "1"        # DC values beside the planes (what the pipeline does)
dcs = (g.coef_shorts // 64 + 127) // 128 * 128
d_dc = lib.DeviceBuffer(dcs * 2 * n)
for rep in range(3):
    t0 = time.perf_counter()
    rounds = hb.decode_split(d_coef.ptr, stride, d_dc.ptr, dcs) if split else hb.decode(d_coef.ptr, stride)
    t_h = time.perf_counter() - t0
    t0 = time.perf_counter()
    lib.check(lib.L.jga_idct_rgb_batch_dc(C.byref(g), n, d_coef.ptr, stride, d_dc.ptr if split else None, dcs,
                                          d_q.ptr, 1, d_rgb.ptr, ostride, None))
    lib.check(lib.L.jga_stream_sync(None))
    t_i = time.perf_counter() - t0
    mp = n * w * h / 1e6
    print("%dx%d %s x%d ri=%d: prepare(host parse+H2D %.1f MB) %.1f ms | huffman %.3f ms (%d rounds, "
          "%.0f Mpix/s) | idct+rgb %.3f ms | device total %.0f Mpix/s" % (
              w, h, samp, n, ri, hb.upload_bytes() / 1e6, t_prep * 1e3, t_h * 1e3, rounds,
              mp / t_h, t_i * 1e3, mp / (t_h + t_i)))
want = lib.entropy_decode(jpegs[0], g)
hb.decode(d_coef.ptr, stride)                      # finished planes for the comparison
got = d_coef.download(g.coef_shorts * 2, dtype=np.int16)
m = lib.real_coef_mask(g)
print("coefficients equal host stage:", bool(np.array_equal(got[m], want[m])))
import oracle  # noqa: E402
print("rgb of image 0 equals oracle (split=%s):" % split,
      bool(np.array_equal(d_rgb.download(g.rgb_bytes), oracle.Oracle().decode_rgb(jpegs[0])[1].reshape(-1))))
