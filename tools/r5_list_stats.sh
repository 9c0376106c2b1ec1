#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for cfg in "3840 2160 420 48 0" "1920 1080 420 1 0" "3840 2160 420 1 0" "1920 1080 420 16 0"; do
  echo "== $cfg"
  env JGA_LIB_PATH=$T JGA_HUFF_LIST_STATS=1 python tools/hbench.py $cfg 2>&1 | grep -E "list round|huffman" | tail -16
done
