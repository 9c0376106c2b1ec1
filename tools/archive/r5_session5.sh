#!/bin/bash
# round 5, session 5: one clear launch instead of the side stream; shard, lone-frame latencies, timeline, GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s5
timeout 300 python tools/shard_sweep.py 128 "" "unstuff=1" "spin_waits=1" > gpurun_out/r5s5/shard.txt 2>&1
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/shard_sweep.py 128 "" "spin_waits=1" >> gpurun_out/r5s5/shard.txt 2>&1
cat gpurun_out/r5s5/shard.txt
timeout 300 bash tools/shard_timeline.sh pinned=1 > gpurun_out/r5s5/timeline.txt 2>&1
cp gpurun_out/stl/*kernel_trace.csv gpurun_out/r5s5/tl_kernels.csv; cp gpurun_out/stl/*memory_copy_trace.csv gpurun_out/r5s5/tl_copies.csv
cat gpurun_out/r5s5/timeline.txt
timeout 600 python - > gpurun_out/r5s5/latency.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from jpeg_gpu_amd import abi, lib, synth
import configs_bench as cb
for name, w, h, samp, ri in (("1080p 4:2:0", 1920, 1080, "420", 0), ("4K 4:2:0", 3840, 2160, "420", 0), ("4K 4:4:4", 3840, 2160, "444", 0), ("8K 4:2:0 DRI", 7680, 4320, "420", -1)):
    f = synth.synthetic_jpeg(w, h, samp, quality=90, seed=1234, restart_interval=ri)
    lat = min(cb._pipeline_latency(lib, abi, f, 8, reps=20) for _ in range(3))
    plug = cb._plugin(lib, abi, f, 20)
    dev = cb._device_only(lib, [f], 1, 8)
    print("%-14s pipeline one frame -> RGB in HBM %.3f ms | plugin decode_image(RGB) -> host pixels %.3f ms | device only %.3f ms" % (name, lat * 1e3, plug["ms_per_frame"], dev["ms"]), flush=True)
PY
cat gpurun_out/r5s5/latency.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5s5/pytest.txt 2>&1
tail -8 gpurun_out/r5s5/pytest.txt
