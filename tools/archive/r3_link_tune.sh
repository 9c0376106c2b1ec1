#!/bin/bash
# with the uploads taking turns: device-turn budget, lanes, group sizes again
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for s in 3 2 4 6; do echo -n "DEVICE_SLOTS=$s :: "; JGA_PIPE_DEVICE_SLOTS=$s SWEEP_CFGS="48,8,24" timeout 300 python tools/e2e_sweep2.py 1536 2>&1 | tail -1; done
SWEEP_CFGS="48,8,24 48,6,24 48,12,24 32,8,24 64,8,24 32,12,24 48,8,16 48,8,32" timeout 900 python tools/e2e_sweep2.py 1536 2>&1 | tail -8
for cfg in "JGA_PIPE_MIN_GROUP=4 JGA_PIPE_GROUPS_PER_LANE=4" "JGA_PIPE_MIN_GROUP=2 JGA_PIPE_GROUPS_PER_LANE=4" "JGA_PIPE_MIN_GROUP=8 JGA_PIPE_GROUPS_PER_LANE=4" "JGA_PIPE_MIN_GROUP=4 JGA_PIPE_GROUPS_PER_LANE=8" "JGA_PIPE_MIN_GROUP=2 JGA_PIPE_GROUPS_PER_LANE=8" "JGA_PIPE_MIN_GROUP=4 JGA_PIPE_GROUPS_PER_LANE=2" "JGA_PIPE_MIN_GROUP=3 JGA_PIPE_GROUPS_PER_LANE=6"; do
  echo -n "$cfg :: "; env $cfg python tools/r3_ramp.py 2>&1 | tail -1
done
