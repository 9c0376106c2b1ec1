"""BASELINE.json configs 2-5 on one MI355X (parity is in tests/; this prints rates).
Usage: python tools/configs_bench.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from jpeg_gpu_amd import abi, lib, synth  # noqa: E402


def device_stage(jpegs, n, reps=20):
    """coefficient planes resident -> RGB (HIP events), and scan bytes resident -> RGB."""
    hdr, g = lib.geom_of(jpegs[0])
    cs = (g.coef_shorts * 2 + 255) // 256 * 128
    os_ = (g.rgb_bytes + 255) // 256 * 256
    dc, do, dq = lib.DeviceBuffer(cs * 2 * n), lib.DeviceBuffer(os_ * n), lib.DeviceBuffer(384 * n)
    for i in range(n):
        dc.upload(lib.entropy_decode(jpegs[i % len(jpegs)], g), offset=i * cs * 2)
    dq.upload(np.tile(lib.qtab_of(hdr).reshape(-1), n))
    ms = C.c_float()
    for r in (3, reps):
        lib.check(lib.L.jga_time_idct_batch(C.byref(g), n, dc.ptr, cs, dq.ptr, 1, do.ptr, os_, 1, r,
                                            None, C.byref(ms)))
    t_k = ms.value * 1e-3
    jobs = [jpegs[i % len(jpegs)] for i in range(n)]
    hb = lib.HuffBatch(n, sum(map(len, jobs)) + 4096 * n)
    hb.prepare(jobs)
    lib.check(lib.L.jga_stream_sync(None))
    dq.upload(hb.qtabs())
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        rounds = hb.decode(dc.ptr, cs)
        lib.check(lib.L.jga_idct_rgb_batch(C.byref(g), n, dc.ptr, cs, dq.ptr, 1, do.ptr, os_, None))
        lib.check(lib.L.jga_stream_sync(None))
        best = min(best, time.perf_counter() - t0)
    hb.close()
    dc.free(); do.free(); dq.free()
    px = n * g.width * g.height
    ab = n * (g.coef_blocks * 128 + g.rgb_bytes)
    return {"kernel_ms": round(t_k * 1e3, 4), "kernel_Gpx_s": round(px / t_k / 1e9, 1),
            "kernel_GBps": round(ab / t_k / 1e9, 1),
            "gpu_entropy_plus_kernel_ms": round(best * 1e3, 3),
            "gpu_entropy_plus_kernel_Gpx_s": round(px / best / 1e9, 2), "sync_rounds": rounds}


def plugin_latency(jpeg, reps=10):
    with lib.Decoder(jpeg) as d:
        d.read_header()
        d.init_image()
        d.decode(abi.JPEG_DECODE_RGB)
        t0 = time.perf_counter()
        for _ in range(reps):
            d.reset()
            d.read_header()
            d.decode(abi.JPEG_DECODE_RGB)
        return (time.perf_counter() - t0) / reps


def pipeline(jobs, transport, **kw):
    pl = lib.Pipeline(device=0, out=abi.JPEG_DECODE_RGB, transport=transport, **kw)
    pl.run(jobs)                                  # warm: every lane sized for its groups
    t0 = time.perf_counter()
    rc, _ = pl.run(jobs)
    dt = time.perf_counter() - t0
    pl.close()
    assert rc == 0
    return dt


def host_entropy_ms(jpeg):
    _, g = lib.geom_of(jpeg)
    lib.entropy_decode(jpeg, g)
    t0 = time.perf_counter()
    lib.entropy_decode(jpeg, g)
    return (time.perf_counter() - t0) * 1e3


print("config 2: 1920x1080 4:2:0 q90, one image")
j = [synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=1234)]
print("  ", device_stage(j, 1), "| plugin decode_image(RGB) incl. D2H %.2f ms" % (plugin_latency(j[0]) * 1e3),
      "| host entropy stage %.1f ms" % host_entropy_ms(j[0]))
print("config 3: 3840x2160 4:4:4 q90, one image")
j = [synth.synthetic_jpeg(3840, 2160, "444", quality=90, seed=1234)]
print("  ", device_stage(j, 1), "| plugin decode_image(RGB) incl. D2H %.2f ms" % (plugin_latency(j[0]) * 1e3),
      "| host entropy stage %.1f ms" % host_entropy_ms(j[0]))
print("config 4: 1024 x 1080p 4:2:0 (seeds 0..15 repeated), ONE GPU takes all 1024 / its 128-image shard")
j = [synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=s) for s in range(16)]
print("   1024 resident:", device_stage(j, 1024, reps=5))
print("    128 resident:", device_stage(j, 128, reps=10))
jobs = [j[i % 16] for i in range(1024)]
px = 1024 * 1920 * 1080
for tr, kw in ((2, dict(nthreads=96, batch=32, depth=4)), (0, dict(nthreads=48))):
    dt = pipeline(jobs, tr, **kw)
    print("   pipeline transport %d: JPEG in host RAM -> RGB in HBM, 1024 images %.1f ms = %.1f Gpixel/s"
          % (tr, dt * 1e3, px / dt / 1e9))
print("config 5: 7680x4320 4:2:0 q90, DRI = one MCU row (480 MCUs)")
j = [synth.synthetic_jpeg(7680, 4320, "420", quality=90, seed=1234, restart_interval=-1)]
print("   1 image :", device_stage(j, 1), "| host entropy stage %.1f ms" % host_entropy_ms(j[0]),
      "| plugin decode_image(RGB) incl. D2H %.2f ms" % (plugin_latency(j[0], 5) * 1e3))
print("   8 images:", device_stage(j, 8, reps=5))
