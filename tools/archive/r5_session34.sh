#!/bin/bash
# round 5, session 34: fuzz + soak again after the 12-bit packs (every single-image decode takes them), the clears
# inside the init launch and the single verdict copy; the new mixed-tables test; soak_gpu_huff
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s34
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mixed_tables or huffman" > gpurun_out/r5s34/pytest.txt 2>&1; tail -3 gpurun_out/r5s34/pytest.txt
( timeout 900 python tools/fuzz_gpu_huff.py 91 6000 ) > gpurun_out/r5s34/fuzz.txt 2>&1; tail -2 gpurun_out/r5s34/fuzz.txt
( timeout 900 python tools/fuzz_gpu_huff.py 92 4000 wide ) >> gpurun_out/r5s34/fuzz.txt 2>&1; tail -2 gpurun_out/r5s34/fuzz.txt
( timeout 600 python tools/soak_pipeline.py 60 11 ) > gpurun_out/r5s34/soak_pipeline.txt 2>&1; tail -5 gpurun_out/r5s34/soak_pipeline.txt
( timeout 600 python tools/soak_gpu_huff.py 60 ) > gpurun_out/r5s34/soak_huff.txt 2>&1; tail -5 gpurun_out/r5s34/soak_huff.txt
( timeout 600 python tools/leak_check.py 8 ) > gpurun_out/r5s34/leak.txt 2>&1; tail -4 gpurun_out/r5s34/leak.txt
