#!/bin/bash
# the write pass one lane per block (small batches) against hj_write (JGA_HUFF_BY_BLOCK=0), tuning build, device only
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for cfg in "1920 1080 420 1 0" "3840 2160 420 1 0" "3840 2160 444 1 0" "1920 1080 420 4 0" "7680 4320 420 1 120" "3840 2160 420 2 0" "1920 1080 grey 1 0"; do
  for pass in 1 2; do
    for v in ${VARIANTS:-default 0 1000000}; do
      if [ "$v" = default ]; then e=""; else e="JGA_HUFF_BY_BLOCK=$v"; fi
      echo "== $cfg | by_block=$v"
      env JGA_LIB_PATH=$T $e python tools/hbench.py $cfg 2>&1 | grep -E "huffman|equal" | tail -3
    done
  done
done
