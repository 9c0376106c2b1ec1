#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s8
echo "== without torch" > gpurun_out/r5s8/dbg.txt
timeout 300 python tools/archive/r5_shard_dbg.py >> gpurun_out/r5s8/dbg.txt 2>&1
echo "== with torch imported first" >> gpurun_out/r5s8/dbg.txt
timeout 300 python tools/archive/r5_shard_dbg.py --torch >> gpurun_out/r5s8/dbg.txt 2>&1
echo "== JGA_PIPE_COPY_WAITING=0" >> gpurun_out/r5s8/dbg.txt
JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so JGA_PIPE_COPY_WAITING=0 timeout 300 python tools/archive/r5_shard_dbg.py >> gpurun_out/r5s8/dbg.txt 2>&1
cat gpurun_out/r5s8/dbg.txt
