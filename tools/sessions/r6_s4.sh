#!/bin/bash
# round 6, session 4: the GPU suite after the pruning (hj_sync_sparse, 256/512-byte subsequences, policy as one table),
# the policy sweep (batch alone / device shared), the default bench line
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_s4; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
export JGA_LIB_PATH=$PWD/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
timeout 1500 python tools/policy_sweep.py > $O/policy_alone.txt 2> $O/policy_alone.err; tail -30 $O/policy_alone.txt
timeout 1500 python tools/policy_sweep.py --shared > $O/policy_shared.txt 2> $O/policy_shared.err; tail -30 $O/policy_shared.txt
unset JGA_LIB_PATH
timeout 900 python bench.py > $O/bench.out 2> $O/bench.err; echo "bench rc=$?"; wc -c $O/bench.out; cp bench_details.json $O/bench_details.json
