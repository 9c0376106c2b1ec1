"""GPU entropy stage on streams that do not self-synchronise (flat / periodic data).
Usage: python tools/periodic_streams.py"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from jpeg_gpu_amd import lib, synth
def run(name, data):
    _, g = lib.geom_of(data)
    hb = lib.HuffBatch(1, len(data) + 4096)
    try:
        hb.prepare([data])
        stride = (g.coef_shorts * 2 + 255) // 256 * 128
        d = lib.DeviceBuffer(stride * 2)
        t0 = time.perf_counter(); rounds = hb.decode(d.ptr, stride); dt = time.perf_counter() - t0
        m = lib.real_coef_mask(g)
        ok = np.array_equal(d.download(g.coef_shorts * 2, dtype=np.int16)[m], lib.entropy_decode(data, g)[m])
        print(name, len(data), "bytes:", rounds, "rounds, %d subsequences walked by the host, %.2f ms, equal %s" % (hb.assisted(), dt * 1e3, ok))
    except Exception as e:
        print(name, "FAILED:", e)
for samp in ("420", "444", "grey"):
    for w, h in ((3840, 2160), (1920, 1080)):
        n = synth.coef_shorts(w, h, samp)
        lv = np.zeros(n, np.int16)
        run("zero %s %dx%d" % (samp, w, h), synth.encode_levels(lv, w, h, samp))
        lv = np.zeros(n, np.int16); lv.reshape(-1, 64)[:, 0] = 5; lv.reshape(-1, 64)[::2, 0] = -5
        run("dc alternating %s %dx%d" % (samp, w, h), synth.encode_levels(lv, w, h, samp))
        lv = np.zeros(n, np.int16); lv.reshape(-1, 64)[:, 1] = 1
        run("one ac %s %dx%d" % (samp, w, h), synth.encode_levels(lv, w, h, samp))
        lv = np.zeros(n, np.int16); lv.reshape(-1, 64)[:, 63] = 1
        run("last ac (ZRLs) %s %dx%d" % (samp, w, h), synth.encode_levels(lv, w, h, samp))
