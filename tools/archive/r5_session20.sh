#!/bin/bash
# round 5, session 20: fuzz + soak + leak check of the restructured entropy stage and pipeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s20
( timeout 900 python tools/fuzz_gpu_huff.py 77 3000 ) > gpurun_out/r5s20/fuzz.txt 2>&1; tail -4 gpurun_out/r5s20/fuzz.txt
( timeout 900 python tools/fuzz_gpu_huff.py 78 2500 wide ) >> gpurun_out/r5s20/fuzz.txt 2>&1; tail -4 gpurun_out/r5s20/fuzz.txt
( timeout 600 python tools/soak_pipeline.py 60 9 ) > gpurun_out/r5s20/soak_pipeline.txt 2>&1; tail -4 gpurun_out/r5s20/soak_pipeline.txt
( timeout 600 python tools/soak_input_cache.py 30 5 ) > gpurun_out/r5s20/soak_cache.txt 2>&1; tail -4 gpurun_out/r5s20/soak_cache.txt
( timeout 600 python tools/leak_check.py 12 ) > gpurun_out/r5s20/leak.txt 2>&1; tail -8 gpurun_out/r5s20/leak.txt
