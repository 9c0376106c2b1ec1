#!/bin/bash
# round 5, session 15: rows kernel, half-plane LDS layout A/B + parity + bank conflicts
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s15
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_rare_sampling.py tests/test_baseline_configs.py -m gpu -x -q -k "kernel or extreme or golden or sampling or config or ieee" > gpurun_out/r5s15/pytest.txt 2>&1
tail -3 gpurun_out/r5s15/pytest.txt
for rep in 1 2 3 4; do for f in jpeg_gpu_amd/variants/rows_*.so; do for s in 444 422 440; do
  JGA_LIB_PATH=$PWD/$f timeout 120 python tools/kbench.py --roofline-leg 3840 2160 $s 24 2>/dev/null | grep RESULT | python -c "
import json,sys; d=json.loads(sys.stdin.read().split('RESULT ')[1]); print('%-28s %s  %.4f ms  %.0f GB/s' % ('$f'.split('/')[-1], '$s', d['ms'], d['gbps']))"
done; done; done | sort > gpurun_out/r5s15/rows_ab.txt
cat gpurun_out/r5s15/rows_ab.txt
bash tools/pmc_kernel.sh jga_idct_rgb_rows r5_pmc_rows2 -- python tools/kbench.py --roofline-leg 3840 2160 444 24 > gpurun_out/r5s15/pmc_rows.txt 2>&1
cat gpurun_out/r5s15/pmc_rows.txt
