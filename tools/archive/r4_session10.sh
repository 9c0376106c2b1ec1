#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - <<PY > gpurun_out/r4s10_plugin.txt 2>&1
import sys, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from jpeg_gpu_amd import abi, lib, synth
import configs_bench
open("/tmp/4k.jpg", "wb").write(synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1234))
open("/tmp/1080p.jpg", "wb").write(synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=1234))
open("/tmp/8k.jpg", "wb").write(synth.synthetic_jpeg(7680, 4320, "420", quality=90, seed=1234, restart_interval=-1))
for mode in (0, 1, -1, -2):
    pc = abi.jga_plugin_config(C.sizeof(abi.jga_plugin_config), mode, 0, 0, 0)
    lib.check(lib.L.jga_plugin_configure(C.byref(pc)))
    for name, w, h, s, ri in (("1080p", 1920, 1080, "420", 0), ("4k", 3840, 2160, "420", 0), ("4k444", 3840, 2160, "444", 0), ("8k_dri", 7680, 4320, "420", -1)):
        data = synth.synthetic_jpeg(w, h, s, quality=90, seed=5, restart_interval=ri)
        print("register_buffers=%2d %-7s %.3f ms/frame" % (mode, name, configs_bench._plugin(lib, abi, data, 12)["ms_per_frame"]), flush=True)
PY
for f in 8k 4k 1080p; do for o in rgb yuv; do for reg in 1 0 -1 -2; do
  echo -n "$f -o $o JPEG_GPU_HIP_REGISTER=$reg: "; JPEG_GPU_HIP_REGISTER=$reg timeout 60 jpeg_gpu_amd/jpeg_gpu_hip -o $o --seconds 2 --check /tmp/$f.jpg 2>&1 | grep FPS | tail -1
done; done; done >> gpurun_out/r4s10_plugin.txt
timeout 600 python tools/shard_sweep.py 128 "" "short_job=3" "short_job=3,input_cache_mb=256" "short_job=3,groups_per_lane=8,min_group=2" > gpurun_out/r4s10_shard.txt 2>&1
for n in 2 4 8 16 48; do echo -n "n=$n "; timeout 100 python tools/kbench.py --roofline-leg 3840 2160 420 $n | tail -1; done > gpurun_out/r4s10_mall.txt 2>&1
cat gpurun_out/r4s10_plugin.txt gpurun_out/r4s10_shard.txt gpurun_out/r4s10_mall.txt
