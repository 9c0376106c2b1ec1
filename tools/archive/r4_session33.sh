#!/bin/bash
# lighter content through the pipeline by the write pass's write-out batch (tuning build, JGA_HUFF_FLUSH; default 16)
# and by group size
cd /root/repo; mkdir -p gpurun_out
for f in 16 8 24 32 48; do echo "JGA_HUFF_FLUSH=$f"; JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so JGA_HUFF_FLUSH=$f python tools/light_sweep.py 2560 "" "batch=48" 2>&1 | grep Gpixel; done
echo "JGA_HUFF_FLUSH=16 again"; JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so JGA_HUFF_FLUSH=16 python tools/light_sweep.py 2560 "" "batch=64" "batch=24" 2>&1 | grep Gpixel
