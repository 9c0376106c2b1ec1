#!/bin/bash
# round 6, session 13: the shard with groups FALLING in size (the last to arrive is small: a short last chain)
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_s13; mkdir -p $O
export JGA_LIB_PATH=$PWD/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
run() { echo "== $*" >> $O/shard.txt; env "$@" timeout 300 python tools/shard_sweep.py 128 "" >> $O/shard.txt 2>&1; }
for rep in 1 2 3; do
  run JGA_PIPE_SHORT_RAMP=0
  run JGA_PIPE_SHORT_RAMP=2
  run JGA_PIPE_SHORT_RAMP=1
done
cat $O/shard.txt
