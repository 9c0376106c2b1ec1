#!/bin/bash
# GPU suite + the stand-alone kernel legs of bench.py, 4:4:4 with both kernels (interleaved).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2k}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for rep in 1 2 3; do for mode in row tile; do
  JGA_RGB444=$mode timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu --no-e2e --no-pack --no-gpu-entropy 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); o=d['other_kernels']
print('444=%-4s 420 %.4f ms (%.0f GB/s)  444 %.4f ms (%.0f GB/s)  yuv %.4f  pass3 %.4f ms (%.0f GB/s)  grey %.4f' % ('$mode', d['roofline']['kernel_ms_per_launch'], d['roofline']['achieved'], o['rgb_444']['ms'], o['rgb_444']['GBps'], o['yuv_stage_420']['ms'], o['yuv_to_rgb_420']['ms'], o['yuv_to_rgb_420']['GBps'], o['grey']['ms']))"
done; done | tee $OUT/kernels.txt
