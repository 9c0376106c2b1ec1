#!/bin/bash
# round 4, GPU session 1: the suite on the refactored library, registration probe, shard sweep, bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r4s1_pytest.txt
timeout 300 python tools/register_probe.py > gpurun_out/r4s1_register.txt 2>&1
timeout 600 python tools/shard_sweep.py > gpurun_out/r4s1_shard.txt 2>&1
timeout 900 python bench.py > gpurun_out/r4s1_bench.json 2> gpurun_out/r4s1_bench.err
tail -3 gpurun_out/r4s1_pytest.txt; cat gpurun_out/r4s1_register.txt; cat gpurun_out/r4s1_shard.txt; tail -5 gpurun_out/r4s1_bench.err
