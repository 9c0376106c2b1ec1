#!/bin/bash
# A/B: list rounds (default) against the sparse / dense later rounds (JGA_HUFF_LIST=0, tuning build), device only
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for cfg in "3840 2160 420 48 0" "1920 1080 420 1 0" "3840 2160 420 1 0" "1920 1080 420 16 0" "1920 1080 420 64 0" "3840 2160 444 24 0" "7680 4320 420 8 120"; do
  for pass in 1 2; do
    for v in ${VARIANTS:-default 0}; do
      if [ "$v" = default ]; then e=""; else e="JGA_HUFF_LIST=$v"; fi
      echo "== $cfg | list=$v"
      env JGA_LIB_PATH=$T $e python tools/hbench.py $cfg 2>&1 | grep -E "huffman|equal" | tail -3
    done
  done
done
