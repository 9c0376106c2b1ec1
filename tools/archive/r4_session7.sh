#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python tools/soak_input_cache.py 40 4 > gpurun_out/r4s7_soak.txt 2>&1
timeout 900 python tools/soak_pipeline.py 150 9 > gpurun_out/r4s7_soak_pipeline.txt 2>&1
tail -12 gpurun_out/r4s7_soak.txt; tail -8 gpurun_out/r4s7_soak_pipeline.txt
