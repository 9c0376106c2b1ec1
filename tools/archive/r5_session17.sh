#!/bin/bash
# round 5, session 17: the shard under the entropy stage's knobs (tuning build): which kernel runs the later rounds, iterations, rounds queued up front
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s17
export JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
run() { echo "== $*" >> gpurun_out/r5s17/shard.txt; env "$@" timeout 300 python tools/shard_sweep.py 128 "" >> gpurun_out/r5s17/shard.txt 2>&1; }
run JGA_NONE=1
run JGA_HUFF_SPARSE_FROM=256
run JGA_HUFF_SPARSE_FROM=2
run JGA_HUFF_ITERS=3,3,4
run JGA_HUFF_ITERS=4,4,4
run JGA_HUFF_ITERS=6,6,3
run JGA_PIPE_MIN_GROUP=2 JGA_PIPE_GROUPS_PER_LANE=2
run JGA_PIPE_MIN_GROUP=3
run JGA_PIPE_LINK_SLOTS=3
run JGA_NONE=2
cat gpurun_out/r5s17/shard.txt
