"""PACK expansion kernel alone: words + block index resident in HBM -> QUANT planes (48 x 4K 4:2:0 q90,
the bench's pack_stage leg).  Usage: python tools/ubench.py [nimages]"""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from jpeg_gpu_amd import lib, synth  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
jpegs = [synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1234 + i) for i in range(4)]
_, g = lib.geom_of(jpegs[0])
pw = [lib.entropy_decode_pack(j, g)[:2] for j in jpegs]
nidx = int(lib.L.jga_index_count(C.byref(g)))
pstride = (max(len(p) for p, _ in pw) + 127) // 128 * 128
hp = np.zeros((B, pstride), np.uint16)
hi = np.zeros((B, nidx), np.int32)
for i in range(B):
    p, ix = pw[i % len(pw)]
    hp[i, :len(p)] = p.view(np.uint16)
    hi[i] = ix
cstride = (g.coef_shorts + 127) // 128 * 128
d_pack, d_idx, d_coef = lib.DeviceBuffer(hp.nbytes), lib.DeviceBuffer(hi.nbytes), lib.DeviceBuffer(cstride * 2 * B)
d_pack.upload(hp)
d_idx.upload(hi)
reps = 20
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.check(lib.L.jga_unpack_batch(C.byref(g), B, d_pack.ptr, pstride, pstride, d_idx.ptr, nidx, d_coef.ptr, cstride, None))
    lib.check(lib.L.jga_stream_sync(None))
    tu = (time.perf_counter() - t0) / reps
nblk = sum(g.plane[p].hblocks * g.plane[p].vblocks for p in range(g.nplanes))
words = sum(len(pw[i % len(pw)][0]) for i in range(B))
ub = words * 2 + B * nblk * (4 + 128)
want = lib.entropy_decode(jpegs[0], g)
m = lib.real_coef_mask(g)
ok = bool(np.array_equal(d_coef.download(g.coef_shorts * 2, dtype=np.int16)[m], want[m]))
print("unpack x%d: %.4f ms  %.0f GB/s = %.3f of 8 TB/s  (%.1f words per block)  equals host stage: %s" % (
    B, tu * 1e3, ub / tu / 1e9, ub / tu / 8e12, words / (B * nblk), ok))
