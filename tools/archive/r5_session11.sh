#!/bin/bash
# round 5, session 11: rows kernel with the wave-uniform in-place clamp: parity + timing; 1080p steady state by clean-up route
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s11
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_rare_sampling.py tests/test_baseline_configs.py -m gpu -x -q -k "kernel or extreme or golden or sampling or config or ieee" > gpurun_out/r5s11/pytest.txt 2>&1
tail -4 gpurun_out/r5s11/pytest.txt
for s in 444 422 440; do
  echo "== 3840 2160 $s" >> gpurun_out/r5s11/kbench.txt
  timeout 200 python tools/kbench.py --roofline-leg 3840 2160 $s 24 >> gpurun_out/r5s11/kbench.txt 2>&1
done
cat gpurun_out/r5s11/kbench.txt | grep -E "==|RESULT"
timeout 600 python tools/steady_sweep.py 1920 1080 420 0 "" "unstuff=2" "unstuff=1" "group=16" "group=64" > gpurun_out/r5s11/steady_1080p.txt 2>&1
cat gpurun_out/r5s11/steady_1080p.txt
