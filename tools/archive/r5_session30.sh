#!/bin/bash
# round 5, session 30: write pass in workgroups of one wave for small batches, clears inside the init launch, one
# verdict copy: entropy parity, then lone-frame latencies against JGA_HUFF_WRITE_BLOCK=512 (tuning build), timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s30
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py tests/test_rare_sampling.py -m gpu -x -q > gpurun_out/r5s30/pytest.txt 2>&1
tail -5 gpurun_out/r5s30/pytest.txt
export JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for rep in 1 2; do for v in 0 512 128; do
echo "== JGA_HUFF_WRITE_BLOCK=$v" >> gpurun_out/r5s30/latency.txt
if [ $v != 0 ]; then export JGA_HUFF_WRITE_BLOCK=$v; else unset JGA_HUFF_WRITE_BLOCK; fi
timeout 600 python - >> gpurun_out/r5s30/latency.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from jpeg_gpu_amd import abi, lib, synth
import configs_bench as cb
for name, w, h, samp, ri, n in (("1080p 4:2:0", 1920, 1080, "420", 0, 1), ("4K 4:2:0", 3840, 2160, "420", 0, 1), ("4K 4:4:4", 3840, 2160, "444", 0, 1), ("8K 4:2:0 DRI", 7680, 4320, "420", -1, 1), ("4 x 4K 4:2:0", 3840, 2160, "420", 0, 4), ("16 x 1080p", 1920, 1080, "420", 0, 16)):
    f = synth.synthetic_jpeg(w, h, samp, quality=90, seed=1234, restart_interval=ri)
    lat = min(cb._pipeline_latency(lib, abi, f, 8, reps=20) for _ in range(3)) if n == 1 else 0.0
    plug = cb._plugin(lib, abi, f, 20) if n == 1 else {"ms_per_frame": 0.0}
    dev = cb._device_only(lib, [f], n, 8)
    print("%-14s pipeline one frame -> RGB in HBM %.3f ms | plugin decode_image(RGB) -> host pixels %.3f ms | device only %.3f ms" % (name, lat * 1e3, plug["ms_per_frame"], dev["ms"]), flush=True)
PY
done; done
cat gpurun_out/r5s30/latency.txt
unset JGA_HUFF_WRITE_BLOCK
bash tools/htimeline.sh 1920 1080 420 1 > gpurun_out/r5s30/timeline_1080.txt 2>&1
cat gpurun_out/r5s30/timeline_1080.txt
timeout 200 python tools/hbench.py 2>&1 | tail -3 > gpurun_out/r5s30/hbench48.txt; cat gpurun_out/r5s30/hbench48.txt
