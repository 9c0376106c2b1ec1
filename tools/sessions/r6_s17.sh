#!/bin/bash
# round 6, session 17: PACK expansion, three-sweep trip: 8 / 16 / 32 words per trip, 16-byte aligned block buffers
cd "$(dirname "$0")/../.."
for pass in 1 2 3; do
  for f in jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so jpeg_gpu_amd/variants/packold.so jpeg_gpu_amd/variants/pack36.so jpeg_gpu_amd/variants/packc8.so jpeg_gpu_amd/variants/packc32.so jpeg_gpu_amd/variants/pack36c8.so; do
    echo -n "$(basename $f): "; JGA_LIB_PATH=$PWD/$f timeout 200 python tools/ubench.py 2>&1 | grep unpack
  done
done
