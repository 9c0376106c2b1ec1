#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for f in jpeg_gpu_amd/variants/*.so; do
  echo -n "$(basename $f) "; JGA_LIB_PATH=$PWD/$f timeout 200 python tools/e2e_trace.py 2304 32 8 24 2>&1 | tail -1
done; done
