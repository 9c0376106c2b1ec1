#!/bin/bash
# round 4, GPU session 2: the new pieces tests first, then the whole suite, then the shard sweep with short_job = 2
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pieces or input_cache or unstuff_modes" 2>&1 | tail -25) > gpurun_out/r4s2_new.txt
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25) > gpurun_out/r4s2_pytest.txt
timeout 600 python tools/shard_sweep.py 128 "" "short_job=2" "short_job=2,spin_waits=1" "short_job=2,unstuff=2" "unstuff=2" > gpurun_out/r4s2_shard.txt 2>&1
cat gpurun_out/r4s2_new.txt | tail -12; tail -8 gpurun_out/r4s2_pytest.txt; cat gpurun_out/r4s2_shard.txt
