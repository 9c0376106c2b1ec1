#!/bin/bash
# in-group iterations of the first (dense) round with list rounds behind it: device only
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for cfg in "3840 2160 420 48 0" "1920 1080 420 64 0"; do
  for pass in 1 2; do
    for v in ${VARIANTS:-3,3,6 4,3,6 5,3,6 6,3,6 8,3,6}; do
      echo "== $cfg | iters=$v"
      env JGA_LIB_PATH=$T JGA_HUFF_ITERS=$v python tools/hbench.py $cfg 2>&1 | grep -E "huffman|equal" | tail -3
    done
  done
done
