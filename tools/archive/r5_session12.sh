#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s12
timeout 600 python tools/archive/r5_steady_trace.py > gpurun_out/r5s12/steady_trace.txt 2>&1
cat gpurun_out/r5s12/steady_trace.txt
