#!/bin/bash
# A/B of library variants in jpeg_gpu_amd/variants/*.so on one box (interleaved repeats)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5; do for f in jpeg_gpu_amd/variants/*.so; do
  JGA_LIB_PATH=$PWD/$f python bench.py --no-cpu --no-e2e --no-pack --no-gpu-entropy --steps ${STEPS:-100} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); o=d['other_kernels']
print('%-20s 420 %.4f ms  444 %.4f ms  yuv %.4f ms  grey %.4f ms' % ('$f'.split('/')[-1], d['roofline']['kernel_ms_per_launch'], o['rgb_444']['ms'], o['yuv_stage_420']['ms'], o['grey']['ms']))"
done; done | sort
