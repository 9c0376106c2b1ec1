#!/bin/bash
# round 5, session 29: config 4's shard leg between the other legs of tools/configs_bench.py, one process
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s29
timeout 900 python tools/archive/r5_shard_aging.py > gpurun_out/r5s29/aging.txt 2>&1
cat gpurun_out/r5s29/aging.txt
