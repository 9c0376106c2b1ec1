#!/bin/bash
# where lists start to pay for a batch alone on the device: default (dense later rounds up to 200 k subsequences) against JGA_HUFF_LIST=1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for cfg in "1920 1080 420 8 0" "1920 1080 420 16 0" "1920 1080 420 32 0" "3840 2160 420 4 0" "3840 2160 420 8 0" "3840 2160 444 4 0"; do
  for pass in 1 2; do
    for v in default 1; do
      if [ "$v" = default ]; then e=""; else e="JGA_HUFF_LIST=$v"; fi
      echo "== $cfg | list=$v"
      env JGA_LIB_PATH=$T $e python tools/hbench.py $cfg 2>&1 | grep -E "huffman|equal" | tail -3
    done
  done
done
