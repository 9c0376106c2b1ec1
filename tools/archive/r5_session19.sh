#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s19
JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so timeout 600 python tools/archive/r5_shard_dbg2.py > gpurun_out/r5s19/dbg2.txt 2>&1
cat gpurun_out/r5s19/dbg2.txt
