#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/shard_sweep.py 128 "" "short_job=2" "short_job=2,input_cache_mb=256" "short_job=2,spin_waits=1,input_cache_mb=256" "short_job=2,unstuff=1" > gpurun_out/r4s4_shard.txt 2>&1
timeout 300 python tools/shard_trace.py 128 short_job=2 input_cache_mb=256 2>&1 | awk '/==== traced/{f=1} f' > gpurun_out/r4s4_trace_short.txt
timeout 300 python tools/shard_trace.py 128 short_job=2 pinned=1 2>&1 | awk '/==== traced/{f=1} f' > gpurun_out/r4s4_trace_short_pinned.txt
cat gpurun_out/r4s4_shard.txt; tail -12 gpurun_out/r4s4_trace_short.txt; tail -12 gpurun_out/r4s4_trace_short_pinned.txt
