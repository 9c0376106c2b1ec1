#!/usr/bin/env python3
"""VERDICT r5 item 5, measured instead of priced: config 5's 8K frame (an interval per MCU row) decoded to HOST pixels
 (a) whole — one upload, one entropy decode, one block decode, one copy back (what the plugin's decode_image does), and
 (b) in N bands of MCU rows, each band a decode of its own on its own stream with its own copy back, so that band k + 1
     decodes while band k's rows cross back.
(b) is the BEST CASE of a banded decode_image: the bands' files are cut BEFORE the clock starts (a real call would pay
the marker pass and the copies — ~0.4 ms for this frame, tools/configs_bench.py `host_cut_ms` — inside it), they lie in
pinned memory, and so do the pixels.  Every variant's pixels are compared with the oracle's.

    python tools/band_overlap_probe.py [bands ...]        (default: 2 4 8)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from jpeg_gpu_amd import lib, synth  # noqa: E402
import oracle  # noqa: E402

W, H = 7680, 4320
frame = synth.synthetic_jpeg(W, H, "420", quality=90, restart_interval=-1, seed=1234)
want = oracle.Oracle().decode_rgb(frame)[1].reshape(-1)
host = lib.PinnedBytes(bytes(W * H * 3))                      # the caller's pixels (pinned: D2H at the link's rate)


class Part:
    """One decode of its own: a file in pinned memory, a batch object, a stream, planes + DC array; pixels go to
    d_out + y0*W*3 and from there to host + y0*W*3."""

    def __init__(self, data, y0, d_out):
        self.pin = lib.PinnedBytes(data)
        self.n = len(data)
        self.y0 = y0
        self.stream = lib.L.jga_stream_create()
        self.hb = lib.HuffBatch(1, self.n + 4096)
        lib.L.jga_huff_set_device_unstuff(self.hb.ptr, 1)     # (as the plugin does for a file it may read where it lies)
        lib.L.jga_huff_set_inputs_pinned(self.hb.ptr, 1)
        lib.L.jga_huff_set_threads(self.hb.ptr, 1)
        self.g = self.hb.prepare_at([self.pin.array.ctypes.data], [self.n], stream=self.stream)
        lib.check(lib.L.jga_stream_sync(self.stream))
        g = self.g
        self.cs = (g.coef_shorts * 2 + 255) // 256 * 128
        self.dcs = (g.coef_shorts // 64 + 127) // 128 * 128
        self.d_coef = lib.DeviceBuffer(self.cs * 2)
        self.d_dc = lib.DeviceBuffer(self.dcs * 2)
        self.dst = d_out.ptr + y0 * W * 3

    def queue(self):
        g = self.g
        self.hb.prepare_at([self.pin.array.ctypes.data], [self.n], stream=self.stream)
        self.hb.decode_split_begin(self.d_coef.ptr, self.cs, self.d_dc.ptr, self.dcs, stream=self.stream)
        lib.check(lib.L.jga_idct_rgb_batch_dc(C.byref(g), 1, self.d_coef.ptr, self.cs, self.d_dc.ptr, self.dcs,
                                              self.hb.qtabs_device(), 1, self.dst, g.rgb_bytes, self.stream))
        lib.check(lib.L.jga_memcpy_d2h(host.array.ctypes.data + self.y0 * W * 3, self.dst, g.rgb_bytes, self.stream))

    def finish(self):
        rounds, valid = self.hb.decode_split_end()
        if not valid:                                          # (the speculation did not hold: queue the tail again)
            g = self.g
            lib.check(lib.L.jga_idct_rgb_batch_dc(C.byref(g), 1, self.d_coef.ptr, self.cs, self.d_dc.ptr, self.dcs,
                                                  self.hb.qtabs_device(), 1, self.dst, g.rgb_bytes, self.stream))
            lib.check(lib.L.jga_memcpy_d2h(host.array.ctypes.data + self.y0 * W * 3, self.dst, g.rgb_bytes, self.stream))
            lib.check(lib.L.jga_stream_sync(self.stream))


def measure(parts, reps=12):
    ts = []
    for rep in range(reps + 3):
        host.array[:] = 0
        t0 = time.perf_counter()
        for p in parts:
            p.queue()
        for p in parts:
            p.finish()
        dt = time.perf_counter() - t0
        if rep >= 3:
            ts.append(dt)
        assert np.array_equal(host.array, want), "pixels differ from the oracle"
    ts.sort()
    return ts[0] * 1e3, ts[len(ts) // 2] * 1e3


lib.check(lib.L.jga_set_device(0))
d_out = lib.DeviceBuffer(W * H * 3 + 256)
whole = [Part(frame, 0, d_out)]
print("8K 4:2:0 q90, an interval per MCU row, %.1f MB; to PINNED host pixels (99.5 MB); min / median ms of 12, every run verified" % (len(frame) / 1e6))
print("whole frame, one decode:            %.3f / %.3f ms" % measure(whole))
for n in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
    t0 = time.perf_counter()
    bands = lib.band_plan(frame, n)
    files = [lib.band_file(frame, b) for b in bands]
    cut = (time.perf_counter() - t0) * 1e3
    parts = [Part(f, b.y0, d_out) for f, b in zip(files, bands)]
    lo, med = measure(parts)
    print("%d bands (%s rows), cut outside the clock (%.2f ms on the host): %.3f / %.3f ms" % (
        len(bands), "/".join(str(b.rows) for b in bands[:3]) + ("/..." if len(bands) > 3 else ""), cut, lo, med))
print("(D2H of 99.5 MB alone at 55 GB/s: 1.81 ms)")
