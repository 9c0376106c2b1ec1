"""Per-frame times of the plugin's decode_image(RGB) loop (no output between frames) and, under rocprofv3, the device's
timeline of its last frames.  Usage: python tools/archive/r5_plugin_frames.py <4k|444|1080> [gap: none|write|sleep0]"""
import sys, time, os
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from jpeg_gpu_amd import abi, lib, synth
FILES = {"1080": (1920, 1080, "420"), "4k": (3840, 2160, "420"), "444": (3840, 2160, "444")}
w, h, s = FILES[sys.argv[1]]
gap = sys.argv[2] if len(sys.argv) > 2 else "none"
f = synth.synthetic_jpeg(w, h, s, quality=90, seed=1234)
devnull = open("/dev/null", "w")
with lib.Decoder(f) as d:
    d.read_header(); d.init_image(); d.decode(abi.JPEG_DECODE_RGB)
    ts = []
    for k in range(30):
        t0 = time.perf_counter(); d.reset(); d.read_header(); d.decode(abi.JPEG_DECODE_RGB); ts.append(time.perf_counter() - t0)
        if gap == "write":
            devnull.write("x"); devnull.flush()
        elif gap == "sleep0":
            time.sleep(0)
print(sys.argv[1], gap, " ".join("%.2f" % (t * 1e3) for t in ts))
