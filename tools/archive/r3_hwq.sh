#!/bin/bash
# Does the number of hardware queues the HIP runtime maps the lanes' streams onto matter?
# (GPU_MAX_HW_QUEUES, default 4: eight lane streams share four queues.)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3_hwq.txt; : > $OUT
for q in 4 8 16 2; do
  echo -n "GPU_MAX_HW_QUEUES=$q light/bench :: " | tee -a $OUT; GPU_MAX_HW_QUEUES=$q timeout 300 python tools/r3_light_e2e.py 2>&1 | tail -1 | tee -a $OUT
  echo -n "GPU_MAX_HW_QUEUES=$q :: " | tee -a $OUT; GPU_MAX_HW_QUEUES=$q timeout 300 python tools/r3_pipe_sweep.py 2>&1 | tail -1 | tee -a $OUT
done
