"""Batches with randomly damaged members through the GPU entropy stage with the upload in pieces
(JGA_HUFF_OPT_PIECES) against the same files decoded one by one without pieces: the verdict of every member
(prepare rejects / decode flags / fine) and the planes of every member that is fine must be the same — and
nothing may hang.  Usage: fuzz_pieces.py [seed] [n_batches]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from jpeg_gpu_amd import lib, synth
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 150
stats = dict(batches=0, members=0, damaged=0, prepare_rejected=0, decode_flagged=0, planes_equal=0, planes_differ=0, verdict_differs=0)
t0 = time.time()


def single(d, device_unstuff):
    """-> ('reject' | 'flag' | 'ok', planes or None) of one file decoded alone, no pieces."""
    try:
        _, c, _ = lib.gpu_entropy_decode([d], device_unstuff=device_unstuff)
        return "ok", c[0]
    except lib.JgaError as e:
        return ("flag" if "image 0" in str(e) else "reject"), None


for it in range(nb):
    samp = ["420", "444", "grey", "422", "440", "411"][it % 6]
    ri = [0, 3, -1, 11][it % 4]
    w, h = 200 + (it * 37) % 300, 120 + (it * 23) % 200
    n = int(rng.integers(2, 9))
    files = []
    for k in range(n):
        d = bytearray(synth.synthetic_jpeg(w, h, samp, quality=40 + (it * 7 + k * 11) % 55, restart_interval=ri, seed=it * 100 + k))
        if rng.random() < 0.4:
            stats["damaged"] += 1
            sos = d.find(b"\xff\xda")
            for _ in range(int(rng.integers(1, 5))):
                pos = int(rng.integers(sos + 14, len(d) - 2))
                mode = int(rng.integers(0, 4))
                if mode == 0: d[pos] = int(rng.integers(0, 256))
                elif mode == 1: d[pos] ^= 1 << int(rng.integers(0, 8))
                elif mode == 2: del d[pos]
                else: d = d[:pos] + b"\xff\xd9"
        files.append(bytes(d))
    try:
        geoms = [lib.geom_of(f)[1] for f in files]
    except lib.JgaError:
        continue
    for device_unstuff in (False, True):
        want = [single(f, device_unstuff) for f in files]
        hb = lib.HuffBatch(n, sum(map(len, files)) + 4096 * n, device_unstuff)
        hb.set_option(4, int(rng.integers(2, 6)))            # JGA_HUFF_OPT_PIECES
        stats["batches"] += 1
        stats["members"] += n
        try:
            try:
                g = hb.prepare(files)
            except lib.JgaError:
                got = ["reject" if lib.L.jga_huff_prepare_verdict(hb.ptr, i) else "?" for i in range(n)]
                stats["prepare_rejected"] += got.count("reject")
                for i in range(n):
                    if (got[i] == "reject") != (want[i][0] == "reject"):
                        stats["verdict_differs"] += 1
                continue
            if any(v == "reject" for v, _ in want):
                stats["verdict_differs"] += 1
            stride = lib._align(g.coef_shorts * 2) // 2
            dbuf = lib.DeviceBuffer(stride * 2 * n)
            dbuf.upload(np.full(stride * n, 0x5A5A, np.int16))
            try:
                hb.decode(dbuf.ptr, stride)
            except lib.JgaError:
                pass
            flags = [int(lib.L.jga_huff_image_error(hb.ptr, i) != 0) for i in range(n)]
            raw = dbuf.download(dtype=np.int16).reshape(n, stride)[:, :g.coef_shorts]
            m = lib.real_coef_mask(g)
            for i in range(n):
                v = want[i][0]
                if flags[i]:
                    stats["decode_flagged"] += 1
                    if v != "flag":
                        stats["verdict_differs"] += 1
                elif v != "ok":
                    stats["verdict_differs"] += 1
                elif np.array_equal(raw[i][m], want[i][1][m]):
                    stats["planes_equal"] += 1
                else:
                    stats["planes_differ"] += 1
            dbuf.free()
        finally:
            hb.close()
print(stats, "%.1f s" % (time.time() - t0))
sys.exit(1 if stats["planes_differ"] or stats["verdict_differs"] else 0)
