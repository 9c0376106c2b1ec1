#!/bin/bash
# round 5, session 6: half-line HBM access; lone-frame kernel timelines; the plugin's 4K frame per iteration
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s6
timeout 300 tools/bin/r5_probes halfline > gpurun_out/r5s6/halfline.txt 2>&1
cat gpurun_out/r5s6/halfline.txt
timeout 300 bash tools/htimeline.sh 1920 1080 420 1 0 > gpurun_out/r5s6/htl_1080p.txt 2>&1
cat gpurun_out/r5s6/htl_1080p.txt
timeout 300 bash tools/htimeline.sh 3840 2160 420 1 0 > gpurun_out/r5s6/htl_4k.txt 2>&1
cat gpurun_out/r5s6/htl_4k.txt
timeout 300 python - > gpurun_out/r5s6/plugin_iters.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, ".")
from jpeg_gpu_amd import abi, lib, synth
for name, w, h, samp in (("4K 4:2:0", 3840, 2160, "420"), ("4K 4:4:4", 3840, 2160, "444"), ("1080p", 1920, 1080, "420")):
    f = synth.synthetic_jpeg(w, h, samp, quality=90, seed=1234)
    with lib.Decoder(f) as d:
        d.read_header(); d.init_image(); d.decode(abi.JPEG_DECODE_RGB)
        ts = []
        for _ in range(40):
            t0 = time.perf_counter()
            d.reset(); d.read_header(); d.decode(abi.JPEG_DECODE_RGB)
            ts.append((time.perf_counter() - t0) * 1e3)
    print(name, "file %d bytes" % len(f), " ".join("%.2f" % t for t in ts), flush=True)
PY
cat gpurun_out/r5s6/plugin_iters.txt
