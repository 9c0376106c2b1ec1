"""Kernel micro-benchmark: time the fused RGB / YUV kernels on a resident batch
for each load mode (JGA_LOADMODE is read once per process, so each mode runs in
a subprocess).  Usage: python tools/kbench.py [W H sampling nimages]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def roofline_leg(w, h, samp, n, prewarm=0.5, reps=50):
    """bench.py's roofline leg as a program of its own (for rocprofv3): the fused RGB kernel on n resident frames,
    `prewarm` seconds of untimed launches (the clocks settle: the first launches of a fresh process run boosted,
    the next ~20 slow down while the power management catches up), then `reps` launches between two HIP events.
    The LAST `reps` launches of the kernel trace are the timed ones."""
    import ctypes as C
    import time
    import numpy as np
    from jpeg_gpu_amd import lib, synth
    data = synth.synthetic_jpeg(w, h, samp, quality=90, seed=1234)
    hd, g = lib.geom_of(data)
    coef = lib.entropy_decode(data, g)
    cstride = (g.coef_shorts * 2 + 255) // 256 * 128
    ostride = (g.rgb_bytes + 255) // 256 * 256
    d_coef, d_q, d_out = lib.DeviceBuffer(cstride * 2 * n), lib.DeviceBuffer(384 * n), lib.DeviceBuffer(ostride * n)
    for i in range(n):
        d_coef.upload(coef, offset=i * cstride * 2)
    d_q.upload(np.tile(lib.qtab_of(hd).reshape(-1), n))
    stream = lib.L.jga_stream_create()
    ms = C.c_float()
    launch = lambda r: lib.check(lib.L.jga_time_idct_batch(C.byref(g), n, d_coef.ptr, cstride, d_q.ptr, 1, d_out.ptr,
                                                           ostride, 1, r, stream, C.byref(ms)))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < prewarm:
        launch(20)
    launch(5)
    launch(reps)
    alg = (g.coef_blocks * 128 + g.rgb_bytes) * n
    return dict(ms=ms.value, gbps=alg / ms.value / 1e6, reps=reps, algorithmic_bytes_per_launch=alg)


def one(w, h, samp, n, reps=20):
    import ctypes as C
    import numpy as np
    from jpeg_gpu_amd import lib, synth
    data = synth.synthetic_jpeg(w, h, samp, quality=90, seed=1234)
    hd, g = lib.geom_of(data)
    coef = lib.entropy_decode(data, g)
    q = lib.qtab_of(hd)
    cstride = (g.coef_shorts * 2 + 255) // 256 * 128
    res = {}
    d_coef = lib.DeviceBuffer(cstride * 2 * n)
    d_q = lib.DeviceBuffer(3 * 64 * 2 * n)
    for i in range(n):
        d_coef.upload(coef, offset=i * cstride * 2)
    d_q.upload(np.tile(q.reshape(-1), n))
    for rgb in (1, 0):
        ob = g.rgb_bytes if rgb else g.yuv_bytes
        ostride = (ob + 255) // 256 * 256
        d_out = lib.DeviceBuffer(ostride * n)
        ms = C.c_float()
        for _ in range(2):
            lib.check(lib.L.jga_time_idct_batch(C.byref(g), n, d_coef.ptr, cstride, d_q.ptr, 1,
                                                d_out.ptr, ostride, rgb, reps, None,
                                                C.byref(ms)))
        alg = (g.coef_blocks * 128 + ob) * n
        res["rgb" if rgb else "yuv"] = dict(ms=ms.value, gbps=alg / ms.value / 1e6,
                                            mpix=w * h * n / ms.value / 1e3)
        d_out.free()
    return res


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--roofline-leg":
        w, h, samp, n = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
        print("RESULT " + json.dumps(roofline_leg(w, h, samp, n)))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        w, h, samp, n = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
        print("RESULT " + json.dumps(one(w, h, samp, n)))
        sys.exit(0)
    args = sys.argv[1:] or ["3840", "2160", "420", "32"]
    configs = [dict(JGA_STAGED="1"), dict(JGA_STAGED="0")]
    if os.environ.get("KBENCH_CONFIGS"):
        configs = json.loads(os.environ["KBENCH_CONFIGS"])
    for cfg in configs:
        env = dict(os.environ, **cfg)
        r = subprocess.run([sys.executable, __file__, "--child"] + args, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if line:
            d = json.loads(line[0][7:])
            print("%-70s rgb %.4f ms %5.0f GB/s | yuv %.4f ms %5.0f GB/s" % (
                json.dumps(cfg).replace("JGA_", ""), d["rgb"]["ms"], d["rgb"]["gbps"],
                d["yuv"]["ms"], d["yuv"]["gbps"]))
        else:
            print(cfg, r.stdout[-1500:])
