#!/bin/bash
# round 5, session 16: tile kernel (4:2:0 headline, 4:1:1), half-plane chroma hand-off: parity + A/B + conflicts
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s16
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_rare_sampling.py tests/test_baseline_configs.py -m gpu -x -q -k "kernel or extreme or golden or sampling or config or ieee or odd or sizes" > gpurun_out/r5s16/pytest.txt 2>&1
tail -3 gpurun_out/r5s16/pytest.txt
for rep in 1 2 3 4; do for f in jpeg_gpu_amd/variants/rows_halfplanes.so jpeg_gpu_amd/variants/tile_halfplanes.so; do for s in 420 411; do
  JGA_LIB_PATH=$PWD/$f timeout 120 python tools/kbench.py --roofline-leg 3840 2160 $s 48 2>/dev/null | grep RESULT | python -c "
import json,sys; d=json.loads(sys.stdin.read().split('RESULT ')[1]); print('%-28s %s  %.4f ms  %.0f GB/s' % ('$f'.split('/')[-1], '$s', d['ms'], d['gbps']))"
done; done; done | sort > gpurun_out/r5s16/tile_ab.txt
cat gpurun_out/r5s16/tile_ab.txt
bash tools/pmc_kernel.sh jga_idct_rgb_kernel r5_pmc_rgb2 -- python tools/kbench.py --roofline-leg 3840 2160 420 48 > gpurun_out/r5s16/pmc_rgb420.txt 2>&1
cat gpurun_out/r5s16/pmc_rgb420.txt
