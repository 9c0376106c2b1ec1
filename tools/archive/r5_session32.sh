#!/bin/bash
# round 5, session 32: hipMemcpyBatchAsync against a loop of copies (16 x 0.78 MB, one stream)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s32
timeout 300 python tools/batch_copy_probe.py > gpurun_out/r5s32/batch.txt 2>&1
cat gpurun_out/r5s32/batch.txt
