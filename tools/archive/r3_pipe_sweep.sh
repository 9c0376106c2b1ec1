#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3_pipe_sweep3.txt; : > $OUT
for cfg in "JGA_PIPE_GROUPS_PER_LANE=4 JGA_PIPE_MIN_GROUP=4" \
           "JGA_PIPE_GROUPS_PER_LANE=4 JGA_PIPE_MIN_GROUP=2" \
           "JGA_PIPE_GROUPS_PER_LANE=4 JGA_PIPE_MIN_GROUP=8" \
           "JGA_PIPE_GROUPS_PER_LANE=2 JGA_PIPE_MIN_GROUP=4" \
           "JGA_PIPE_GROUPS_PER_LANE=4 JGA_PIPE_MIN_GROUP=4 JGA_PIPE_DEVICE_SLOTS=2" \
           "JGA_PIPE_GROUPS_PER_LANE=4 JGA_PIPE_MIN_GROUP=4 JGA_PIPE_DEVICE_SLOTS=4" \
           "JGA_PIPE_GROUPS_PER_LANE=4 JGA_PIPE_MIN_GROUP=4"; do
  echo -n "$cfg :: " | tee -a $OUT; env $cfg timeout 300 python tools/r3_pipe_sweep.py 2>&1 | tail -1 | tee -a $OUT
done
python tools/r3_trace_shard.py 128 2>&1 | tail -10 | tee -a $OUT
