#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3_knobs.txt; : > $OUT
for cfg in "X=0" "JGA_HUFF_ITERS=2,3,6" "JGA_HUFF_ITERS=4,3,6" "JGA_HUFF_ITERS=5,3,6" "JGA_HUFF_ITERS=3,2,6" "JGA_HUFF_ITERS=3,4,6" "JGA_HUFF_ITERS=4,4,6" "JGA_HUFF_ITERS=6,6,6" \
           "JGA_HUFF_LITE=33" "JGA_HUFF_LITE=65" "JGA_HUFF_LITE=1" "JGA_HUFF_LITE=0" "JGA_HUFF_FLUSH=12" "JGA_HUFF_FLUSH=24" "JGA_HUFF_FLUSH=32" "JGA_HUFF_SPARSE_FROM=2" "X=1"; do
  echo -n "$cfg :: " | tee -a $OUT
  for r in 1 2; do env $cfg timeout 200 python tools/hbench.py 2>&1 | grep "x48" | tail -1 | sed 's/.*huffman \([0-9.]*\) ms (\([0-9]*\) rounds.*/\1ms\/\2r /' | tr -d '\n' | tee -a $OUT; done; echo | tee -a $OUT
done
