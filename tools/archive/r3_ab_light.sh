#!/bin/bash
# lighter + bench content through the pipeline, interleaved over the library builds in jpeg_gpu_amd/variants/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do for f in jpeg_gpu_amd/variants/*.so; do
  echo -n "$(basename $f) :: "; JGA_LIB_PATH=$PWD/$f timeout 300 python tools/r3_light_e2e.py 2>&1 | tail -1
done; done
for f in jpeg_gpu_amd/variants/*.so; do
  echo -n "$(basename $f) :: "; JGA_LIB_PATH=$PWD/$f timeout 200 python tools/hbench.py 2>&1 | grep "Mpix/s" | tail -1
  echo -n "$(basename $f) light :: "; CONTENT=light JGA_LIB_PATH=$PWD/$f timeout 200 python tools/hbench.py 3840 2160 420 32 2>&1 | grep "Mpix/s" | tail -1
done
