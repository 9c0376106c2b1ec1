#!/bin/bash
# round 5, session 35: in-group iterations per launch for lone frames (JGA_HUFF_ITERS=first,later,rounds per host check)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s35
export JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for rep in 1 2; do for it in "" "4,4,6" "6,6,6" "9,9,6" "12,12,4" "6,3,6" "12,3,6" "5,5,4"; do
if [ -n "$it" ]; then export JGA_HUFF_ITERS=$it; else unset JGA_HUFF_ITERS; fi
timeout 300 python - >> gpurun_out/r5s35/iters.txt 2>&1 <<PY
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from jpeg_gpu_amd import abi, lib, synth
import configs_bench as cb
out = []
for name, w, h, samp in (("1080p", 1920, 1080, "420"), ("4K", 3840, 2160, "420"), ("4K444", 3840, 2160, "444")):
    f = synth.synthetic_jpeg(w, h, samp, quality=90, seed=1234)
    lat = min(cb._pipeline_latency(lib, abi, f, 8, reps=20) for _ in range(2))
    dev = cb._device_only(lib, [f], 1, 8)
    out.append("%s %.3f / %.3f (%d rounds)" % (name, lat * 1e3, dev["ms"], dev["sync_rounds"]))
print("iters %-9s  pipeline / device-only ms:  %s" % ("${it:-default}", " | ".join(out)), flush=True)
PY
done; done
cat gpurun_out/r5s35/iters.txt
