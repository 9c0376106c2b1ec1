#!/bin/bash
# round 5, session 1: the probes the short-job work starts from + where the shard stands on this box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s1
timeout 300 tools/bin/r5_probes all > gpurun_out/r5s1/probes.txt 2>&1
timeout 300 python tools/shard_sweep.py 128 "" > gpurun_out/r5s1/shard_default.txt 2>&1
timeout 300 python tools/shard_sweep.py 128 "input_cache_mb=512" "spin_waits=1" >> gpurun_out/r5s1/shard_default.txt 2>&1
timeout 300 bash tools/shard_timeline.sh > gpurun_out/r5s1/timeline_default.txt 2>&1
timeout 300 bash tools/shard_timeline.sh pinned=1 > gpurun_out/r5s1/timeline_pinned.txt 2>&1
JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so timeout 120 python tools/shard_trace.py 128 > gpurun_out/r5s1/trace_default.txt 2>&1
cat gpurun_out/r5s1/probes.txt | tail -120
cat gpurun_out/r5s1/shard_default.txt
