#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2packs}; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q -k "huffman or unstuff or pipeline or harness or config or irregular or extreme or corrupted" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for rep in 1 2 3; do timeout 200 python tools/hbench.py 2>&1 | grep "x48" | tail -1 | sed 's/.*| huffman/huffman/'; done | tee $OUT/hbench.txt
timeout 200 python tools/hbench.py 1920 1080 420 1 2>&1 | grep "x1 " | tail -1 | tee -a $OUT/hbench.txt
timeout 200 python tools/hbench.py 7680 4320 420 8 -1 2>&1 | grep "x8 " | tail -1 | tee -a $OUT/hbench.txt
SWEEP_CFGS="32,8,24" timeout 200 python tools/e2e_sweep2.py 2304 | tee -a $OUT/hbench.txt
