#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python - <<PY
import sys; sys.path.insert(0, ".")
from jpeg_gpu_amd import synth
open("/tmp/4k.jpg", "wb").write(synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1234))
open("/tmp/1080p.jpg", "wb").write(synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=1234))
open("/tmp/8k.jpg", "wb").write(synth.synthetic_jpeg(7680, 4320, "420", quality=90, seed=1234, restart_interval=-1))
PY
for f in 4k 1080p 8k; do for o in rgb yuv; do for reg in 0 1; do
  echo -n "$f -o $o JGA_PLUGIN_REGISTER=$reg: "; JGA_PLUGIN_REGISTER=$reg timeout 60 jpeg_gpu_amd/jpeg_gpu_hip -o $o --seconds 2 --check /tmp/$f.jpg 2>&1 | grep FPS | tail -1
done; done; done
