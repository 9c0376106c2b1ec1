#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/shard_sweep.py 1024 "" "min_group=8" "min_group=16" "groups_per_lane=2" "groups_per_lane=8" "device_slots=4" "device_slots=6" "link_slots=3" "unstuff=2" "unstuff=2,input_cache_mb=256" "min_group=8,unstuff=2,input_cache_mb=256" "spin_waits=1" > gpurun_out/r4s9_1024.txt 2>&1
cat gpurun_out/r4s9_1024.txt
