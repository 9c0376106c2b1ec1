#!/bin/bash
# interleaved A/B of the built library against jpeg_gpu_amd/variants/*.so on the entropy stage (tools/hbench.py), device only
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "${CFGS[@]:-3840 2160 420 48 0}" ; do :; done
for cfg in "3840 2160 420 48 0" "1920 1080 420 1 0" "3840 2160 420 1 0" "1920 1080 420 64 0" "3840 2160 444 24 0"; do
  for pass in 1 2 3; do
    for f in jpeg_gpu_amd/libjpeg_gpu_amd.so jpeg_gpu_amd/variants/*.so; do
      echo "== $cfg | $(basename $f)"
      env JGA_LIB_PATH=$PWD/$f python tools/hbench.py $cfg 2>&1 | grep -E "huffman|equal" | tail -3
    done
  done
done
