#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "unpack" 2>&1 | tail -4) > gpurun_out/r4s8_pack.txt
for rep in 1 2 3; do for f in jpeg_gpu_amd/variants/pack_*.so; do echo -n "$(basename $f): "; JGA_LIB_PATH=$PWD/$f timeout 200 python tools/ubench.py 2>&1 | tail -1; done; done >> gpurun_out/r4s8_pack.txt
cat gpurun_out/r4s8_pack.txt
