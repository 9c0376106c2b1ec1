#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "input_cache" 2>&1 | tail -3) > gpurun_out/r4s13.txt
timeout 1500 python tools/fuzz_pieces.py 3 250 >> gpurun_out/r4s13.txt 2>&1
timeout 900 python tools/fuzz_gpu_huff.py 11 1500 wide >> gpurun_out/r4s13.txt 2>&1
tail -6 gpurun_out/r4s13.txt
