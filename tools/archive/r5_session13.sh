#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s13
timeout 900 python tools/steady_sweep.py 1920 1080 420 0 "seconds=0.15" "seconds=0.15,keep=0" "seconds=1.0,keep=0" "seconds=1.0" "seconds=1.0,group=16" "seconds=1.0,files=4" > gpurun_out/r5s13/steady.txt 2>&1
cat gpurun_out/r5s13/steady.txt
