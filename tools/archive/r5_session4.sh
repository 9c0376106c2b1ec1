#!/bin/bash
# round 5, session 4: cross-stream dependency latency; the shard with FIFO copy streams against turns; timelines; GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s4
timeout 300 tools/bin/r5_probes xstream > gpurun_out/r5s4/xstream.txt 2>&1
cat gpurun_out/r5s4/xstream.txt
timeout 300 python tools/shard_sweep.py 128 "" "unstuff=1" "spin_waits=1" "input_cache_mb=-1" > gpurun_out/r5s4/shard.txt 2>&1
export JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
echo "== JGA_PIPE_FIFO=0" >> gpurun_out/r5s4/shard.txt
JGA_PIPE_FIFO=0 timeout 300 python tools/shard_sweep.py 128 "" >> gpurun_out/r5s4/shard.txt 2>&1
echo "== GPU_MAX_HW_QUEUES=8" >> gpurun_out/r5s4/shard.txt
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/shard_sweep.py 128 "" >> gpurun_out/r5s4/shard.txt 2>&1
unset JGA_LIB_PATH
cat gpurun_out/r5s4/shard.txt
timeout 300 bash tools/shard_timeline.sh > gpurun_out/r5s4/timeline.txt 2>&1
cp gpurun_out/stl/*kernel_trace.csv gpurun_out/r5s4/tl_kernels.csv; cp gpurun_out/stl/*memory_copy_trace.csv gpurun_out/r5s4/tl_copies.csv
cat gpurun_out/r5s4/timeline.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5s4/pytest.txt 2>&1
tail -15 gpurun_out/r5s4/pytest.txt
