#!/bin/bash
# round 5, session 21: a short run's last groups with the lone batch's kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s21
export JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
run() { echo "== $*" >> gpurun_out/r5s21/shard.txt; env "$@" timeout 300 python tools/shard_sweep.py 128 "" >> gpurun_out/r5s21/shard.txt 2>&1; }
for rep in 1 2; do
run JGA_PIPE_TAIL_ALONE=0
run JGA_PIPE_TAIL_ALONE=1
run JGA_PIPE_TAIL_ALONE=2
run JGA_PIPE_TAIL_ALONE=4
run JGA_PIPE_TAIL_ALONE=8
done
cat gpurun_out/r5s21/shard.txt
