#!/bin/bash
# Uploads one (or two) at a time, in the order the lanes got through prepare (JGA_PIPE_LINK_SLOTS), against all at once
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do for s in ${SLOTS:-0 1 2}; do
  echo -n "LINK_SLOTS=$s pageable :: "; JGA_PIPE_LINK_SLOTS=$s SWEEP_CFGS="48,8,24" timeout 300 python tools/e2e_sweep2.py 1536 2>&1 | tail -1
  echo -n "LINK_SLOTS=$s pinned, device clean-up :: "; JGA_PIPE_LINK_SLOTS=$s PINNED=1 UNSTUFF=2 SWEEP_CFGS="48,8,24" timeout 300 python tools/e2e_sweep2.py 1536 2>&1 | tail -1
  echo -n "LINK_SLOTS=$s :: "; JGA_PIPE_LINK_SLOTS=$s python tools/r3_ramp.py 2>&1 | tail -1
  echo -n "LINK_SLOTS=$s :: "; JGA_PIPE_LINK_SLOTS=$s timeout 300 python tools/r3_light_e2e.py 2>&1 | tail -1
done; done
