#!/bin/bash
# round 5, session 23: 12-bit AC packs for small batches (hj_ltables_wide): parity of the entropy tests, then lone-frame
# latencies with them (default) and without (JGA_HUFF_NO_WIDE=1, tuning build)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s23
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py tests/test_rare_sampling.py -m gpu -x -q > gpurun_out/r5s23/pytest.txt 2>&1
tail -5 gpurun_out/r5s23/pytest.txt
export JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for rep in 1 2; do for v in 0 1; do
echo "== JGA_HUFF_NO_WIDE=$v" >> gpurun_out/r5s23/latency.txt
if [ $v = 1 ]; then export JGA_HUFF_NO_WIDE=1; else unset JGA_HUFF_NO_WIDE; fi
timeout 600 python - >> gpurun_out/r5s23/latency.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from jpeg_gpu_amd import abi, lib, synth
import configs_bench as cb
for name, w, h, samp, ri in (("1080p 4:2:0", 1920, 1080, "420", 0), ("4K 4:2:0", 3840, 2160, "420", 0), ("4K 4:4:4", 3840, 2160, "444", 0), ("1080p 4:4:4", 1920, 1080, "444", 0), ("8K 4:2:0 DRI", 7680, 4320, "420", -1)):
    f = synth.synthetic_jpeg(w, h, samp, quality=90, seed=1234, restart_interval=ri)
    lat = min(cb._pipeline_latency(lib, abi, f, 8, reps=20) for _ in range(3))
    plug = cb._plugin(lib, abi, f, 20)
    dev = cb._device_only(lib, [f], 1, 8)
    print("%-14s pipeline one frame -> RGB in HBM %.3f ms | plugin decode_image(RGB) -> host pixels %.3f ms | device only %.3f ms" % (name, lat * 1e3, plug["ms_per_frame"], dev["ms"]), flush=True)
PY
done; done
cat gpurun_out/r5s23/latency.txt
unset JGA_HUFF_NO_WIDE
timeout 200 python tools/hbench.py 1920 1080 420 1 2>&1 | tail -3 > gpurun_out/r5s23/hbench.txt
JGA_HUFF_NO_WIDE=1 timeout 200 python tools/hbench.py 1920 1080 420 1 2>&1 | tail -3 >> gpurun_out/r5s23/hbench.txt
cat gpurun_out/r5s23/hbench.txt
