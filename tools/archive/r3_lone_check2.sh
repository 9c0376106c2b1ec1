#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py tests/test_harness.py -q -m gpu -x 2>&1 | tail -3
for lean in 0 1; do for geo in "1920 1080 420 1" "3840 2160 420 1" "3840 2160 444 1" "7680 4320 420 1 -1" "3840 2160 420 8" "3840 2160 420 48"; do
  echo -n "lean=$lean $geo :: "; JGA_HUFF_LEAN=$lean timeout 120 python tools/hbench.py $geo 2>&1 | grep "Mpix/s" | tail -1 | sed 's/.*| huffman/huffman/'
done; done
