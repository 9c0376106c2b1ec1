#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -q -m gpu -x 2>&1 | tail -3
for geo in "1920 1080 420 1" "3840 2160 420 1" "3840 2160 444 1" "7680 4320 420 1 -1" "3840 2160 420 8" "3840 2160 420 48"; do
  echo -n "$geo :: "; timeout 120 python tools/hbench.py $geo 2>&1 | grep "Mpix/s" | tail -1 | sed 's/.*| huffman/huffman/'
done
python - <<PY
from jpeg_gpu_amd import synth
open("/tmp/4k.jpg","wb").write(synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1234))
open("/tmp/1080p.jpg","wb").write(synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=1234))
PY
for f in 4k 1080p; do for o in rgb yuv; do for reg in 0 1; do
  echo -n "$f -o $o JGA_PLUGIN_REGISTER=$reg: "; JGA_PLUGIN_REGISTER=$reg timeout 60 jpeg_gpu_amd/jpeg_gpu_hip -o $o --seconds 2 --check /tmp/$f.jpg 2>&1 | grep FPS | tail -1
done; done; done
