#!/bin/bash
# round 5, session 31: a short job's last groups falling in size (..., 12, 8, 5, 3, 2) against eight equal groups:
# config 4's 128-file shard, and 64 / 96 / 32 files, alternating on one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s31
export JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for rep in 1 2 3; do for t in 1 0; do
  echo -n "taper=$t n=128: " >> gpurun_out/r5s31/shard.txt
  JGA_PIPE_TAPER=$t timeout 300 python tools/shard_sweep.py 128 "" 2>&1 | tail -1 >> gpurun_out/r5s31/shard.txt
done; done
for n in 64 96 32 256; do for t in 1 0; do
  echo -n "taper=$t n=$n: " >> gpurun_out/r5s31/shard.txt
  JGA_PIPE_TAPER=$t timeout 300 python tools/shard_sweep.py $n "" 2>&1 | tail -1 >> gpurun_out/r5s31/shard.txt
done; done
cat gpurun_out/r5s31/shard.txt
