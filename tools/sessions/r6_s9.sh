#!/bin/bash
# round 6, session 9: new tests (dense frames settle on the device; photo configs in the bench contract), the persistent
# cache exercised again by the soak
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_s9; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_bench_contract.py tests/test_pipeline_hardening.py -x -q -m gpu -k "dense_frames or contract_keys or configs_process or hardening or input_cache" > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
timeout 600 python tools/soak_input_cache.py 30 6 2>&1 | grep -v amdgpu | tail -8
