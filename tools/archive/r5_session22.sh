#!/bin/bash
# round 5, session 22: the entropy stage on 48 x 4K with the sparse kernel (rows from global memory, one wave per 256
# subsequences, 28 waves per CU) running the FIRST round too, against the dense one (rows in LDS, 12 waves per CU)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s22
export JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for rep in 1 2; do for v in "" 0 1; do
  echo "== JGA_HUFF_SPARSE_FROM=${v:-default}" >> gpurun_out/r5s22/hbench.txt
  if [ -z "$v" ]; then timeout 200 python tools/hbench.py 2>&1 | tail -3 >> gpurun_out/r5s22/hbench.txt
  else JGA_HUFF_SPARSE_FROM=$v timeout 200 python tools/hbench.py 2>&1 | tail -3 >> gpurun_out/r5s22/hbench.txt; fi
done; done
cat gpurun_out/r5s22/hbench.txt
