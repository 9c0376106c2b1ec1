#!/bin/bash
# round 6, session 16: the PACK expansion with its trip in three sweeps (positions, all look-ups, all stores) against
# round 5's branchy loop (variants/packold.so) and the no-scatter probe
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "unpack or pack" 2>&1 | tail -3
for pass in 1 2 3; do
  for f in jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so jpeg_gpu_amd/variants/packold.so jpeg_gpu_amd/variants/packprobe1.so; do
    echo -n "$(basename $f): "; JGA_LIB_PATH=$PWD/$f timeout 200 python tools/ubench.py 2>&1 | grep unpack
  done
done
