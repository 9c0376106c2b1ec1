#!/bin/bash
# round 6, session 15: where the PACK expansion's time is — timing probes (wrong output on purpose): no scatter at all,
# the scatter into one fixed halfword per lane (no bank conflicts), the scatter without the de-zigzag look-up
cd "$(dirname "$0")/../.."
for pass in 1 2 3; do
  for f in jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so jpeg_gpu_amd/variants/packprobe1.so jpeg_gpu_amd/variants/packprobe2.so jpeg_gpu_amd/variants/packprobe3.so; do
    echo -n "$(basename $f): "; JGA_LIB_PATH=$PWD/$f timeout 200 python tools/ubench.py 2>&1 | grep unpack
  done
done
