#!/bin/bash
# GPU suite + the stand-alone kernel legs of bench.py, 4:4:4 with both kernels (interleaved).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2k}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for rep in 1 2 3; do for mode in 1 0; do
  JGA_RGB_ROWS=$mode timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu --no-e2e --no-pack --no-gpu-entropy 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); o=d['other_kernels']
print('rows=$mode 420 %.4f ms (%.0f GB/s)  444 %.4f ms (%.0f GB/s)  422 %.4f ms (%.0f GB/s)' % (d['roofline']['kernel_ms_per_launch'], d['roofline']['achieved'], o['rgb_444']['ms'], o['rgb_444']['GBps'], o['rgb_422']['ms'], o['rgb_422']['GBps']))"
done; done | tee $OUT/kernels.txt
