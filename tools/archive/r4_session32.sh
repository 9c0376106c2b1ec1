#!/bin/bash
# last evidence session of round 4 at HEAD: leaks, soaks, fuzz (the host entropy stage was rewritten; copy kernel and band cut added)
cd /root/repo; mkdir -p gpurun_out
( timeout 900 python tools/leak_check.py 12 2>&1 | tail -14 ) > gpurun_out/r4s32_leak.txt
( timeout 600 python tools/soak_pipeline.py 2>&1 | tail -8 ) > gpurun_out/r4s32_soak_pipeline.txt
( timeout 600 python tools/soak_input_cache.py 2>&1 | tail -8 ) > gpurun_out/r4s32_soak_cache.txt
( timeout 600 python tools/fuzz_pieces.py 5 150 2>&1 | tail -2 ) > gpurun_out/r4s32_fuzz_pieces.txt
( timeout 600 python tools/fuzz_gpu_huff.py 31 2500 wide 2>&1 | tail -1 ) > gpurun_out/r4s32_fuzz_gpu.txt
tail -3 gpurun_out/r4s32_*.txt
