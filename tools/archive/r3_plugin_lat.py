import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jpeg_gpu_amd import abi, lib, synth
for (w, h, s, ri) in ((3840, 2160, "420", 0), (1920, 1080, "420", 0), (3840, 2160, "444", 0), (7680, 4320, "420", -1)):
    f = synth.synthetic_jpeg(w, h, s, quality=90, seed=1234, restart_interval=ri)
    with lib.Decoder(f) as d:
        d.read_header(); d.init_image(); d.decode(abi.JPEG_DECODE_RGB)
        tr = th = td = 0.0
        n = 20
        for _ in range(n):
            t0 = time.perf_counter(); d.reset(); t1 = time.perf_counter(); d.read_header(); t2 = time.perf_counter()
            d.decode(abi.JPEG_DECODE_RGB); t3 = time.perf_counter()
            tr += t1 - t0; th += t2 - t1; td += t3 - t2
        print("%dx%d %s: reset %.3f header %.3f decode_image(RGB) %.3f ms  (file %.2f MB, rgb %.1f MB)" % (
            w, h, s, tr / n * 1e3, th / n * 1e3, td / n * 1e3, len(f) / 1e6, w * h * 3 / 1e6))
        ty = 0.0
        for _ in range(n):
            d.reset(); d.read_header(); t2 = time.perf_counter(); d.decode(abi.JPEG_DECODE_YUV); ty += time.perf_counter() - t2
        print("      decode_image(YUV) %.3f ms" % (ty / n * 1e3))
