#!/bin/bash
# the write pass one lane per block on batches that fill the device: lighter content (40 blocks per subsequence) and the bench's q90 noise (8)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for c in light noise; do
  for cfg in "3840 2160 420 32 0" "1920 1080 420 64 0"; do
    for pass in 1 2; do
      for v in 0 100000000; do
        echo "== $c $cfg | by_block=$v"
        if [ $c = light ]; then export CONTENT=light; else unset CONTENT; fi
        env JGA_LIB_PATH=$T JGA_HUFF_BY_BLOCK=$v python tools/hbench.py $cfg 2>&1 | grep -E "huffman|equal" | tail -3
      done
    done
  done
done
