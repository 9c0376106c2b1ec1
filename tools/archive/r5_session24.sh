#!/bin/bash
# round 5, session 24: where the plugin's +-0.4 ms per 4K frame comes from: D2H copies by buffer kind and NUMA node,
# then decode_image timed on several decoders of one process with the pixel buffer's placement printed
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s24
timeout 300 python tools/d2h_probe.py > gpurun_out/r5s24/d2h.txt 2>&1
cat gpurun_out/r5s24/d2h.txt
cat /proc/sys/kernel/numa_balancing; grep -i "huge\|numa" /proc/meminfo | head
timeout 600 python - > gpurun_out/r5s24/plugin.txt 2>&1 <<'PY'
import sys, time, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from jpeg_gpu_amd import abi, lib, synth
import configs_bench as cb
import d2h_probe as dp
f = synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1234)
ballast = []
for k in range(8):
    with lib.Decoder(f) as d:
        d.read_header(); d.init_image(); d.decode(abi.JPEG_DECODE_RGB)
        p = C.cast(d.img.pixels, C.c_void_p).value
        ts = []
        for _ in range(20):
            t0 = time.perf_counter(); d.reset(); d.read_header(); d.decode(abi.JPEG_DECODE_RGB); ts.append(time.perf_counter() - t0)
        ts.sort()
        print("decoder %d: pixels at %#x (%% 2 MB = %d KB) nodes %s: best %.3f median %.3f ms" % (k, p, (p % (2 << 20)) >> 10, dp.nodes_of(p, 3840*2160*3), ts[0]*1e3, ts[10]*1e3), flush=True)
    ballast.append(bytearray((k + 1) * 1234567))          # change the allocator's history between decoders
PY
cat gpurun_out/r5s24/plugin.txt
