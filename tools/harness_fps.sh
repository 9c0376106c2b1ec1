#!/bin/bash
# FPS of the headless harness on a 4K 4:2:0 q90 file, per -o stage and entropy choice
cd $GRAFT_REPO_ROOT
python - <<PY
from jpeg_gpu_amd import synth
open("/tmp/4k.jpg","wb").write(synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1234))
PY
for e in gpu host; do for o in rgb yuv quant pack; do
  echo "== entropy=$e -o $o"; JPEG_GPU_HIP_ENTROPY=$e ./jpeg_gpu_amd/jpeg_gpu_hip -o $o --seconds 3.2 --check /tmp/4k.jpg | sed -n '2p;$p'   # a full one-second line + the checksum
done; done
