#!/bin/bash
# lighter content through the pipeline against the device-turn budget (after the plane clear went)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for s in 3 2 4 6 3; do
  echo -n "JGA_PIPE_DEVICE_SLOTS=$s :: "; JGA_PIPE_DEVICE_SLOTS=$s timeout 300 python tools/r3_light_e2e.py 2>&1 | tail -1
done
