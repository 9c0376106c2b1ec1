#!/bin/bash
# round 5, session 14: rows kernel A/B on one box (interleaved), 4:4:4 / 4:2:2 / 4:4:0
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s14
for rep in 1 2 3 4; do for f in jpeg_gpu_amd/variants/rows_clamp_*.so; do for s in 444 422 440; do
  JGA_LIB_PATH=$PWD/$f timeout 120 python tools/kbench.py --roofline-leg 3840 2160 $s 24 2>/dev/null | grep RESULT | python -c "
import json,sys; d=json.loads(sys.stdin.read().split('RESULT ')[1]); print('%-28s %s  %.4f ms  %.0f GB/s' % ('$f'.split('/')[-1], '$s', d['ms'], d['gbps']))"
done; done; done | sort > gpurun_out/r5s14/rows_ab.txt
cat gpurun_out/r5s14/rows_ab.txt
