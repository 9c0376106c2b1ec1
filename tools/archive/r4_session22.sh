#!/bin/bash
# band split on the GPU: parity tests, then config 5's frame against its band 3 of 8 (tools/configs_bench.py)
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_band_split.py -x -q 2>&1 | tail -5 > gpurun_out/r4s22_tests.txt
timeout 900 python tools/configs_bench.py > gpurun_out/r4s22_configs.txt 2> gpurun_out/r4s22_configs.err
tail -3 gpurun_out/r4s22_tests.txt; grep config5 gpurun_out/r4s22_configs.txt | cut -c1-3000
