#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2lite}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "huffman or pipeline or harness or config4 or irregular or extreme" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for rep in 1 2 3; do for lite in 1 0; do
  echo -n "lists=$lite "; JGA_HUFF_LISTS=$lite timeout 200 python tools/hbench.py 2>&1 | grep "x48" | tail -1
done; done | tee $OUT/hbench.txt
for lite in 1 0; do echo -n "lists=$lite 1080p x1: "; JGA_HUFF_LISTS=$lite timeout 200 python tools/hbench.py 1920 1080 420 1 2>&1 | grep "x1 " | tail -1; done | tee -a $OUT/hbench.txt
for lite in 1 0; do echo -n "lists=$lite 8K dri x8: "; JGA_HUFF_LISTS=$lite timeout 200 python tools/hbench.py 7680 4320 420 8 -1 2>&1 | grep "x8 " | tail -1; done | tee -a $OUT/hbench.txt
SWEEP_CFGS="32,8,24" python tools/e2e_sweep2.py 2304 | tee -a $OUT/hbench.txt
