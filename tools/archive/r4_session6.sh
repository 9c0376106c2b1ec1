#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for q in 4 16; do
  echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout 300 python tools/shard_sweep.py 128 "" "short_job=2" "short_job=2,spin_waits=1"
done > gpurun_out/r4s6_hwq.txt 2>&1
cat gpurun_out/r4s6_hwq.txt
