#!/bin/bash
# round 5, session 2: the fetch route + one wait per group: GPU suite, the shard, its timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s2
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5s2/pytest.txt 2>&1
tail -15 gpurun_out/r5s2/pytest.txt
timeout 600 python tools/shard_sweep.py 128 > gpurun_out/r5s2/shard.txt 2>&1
cat gpurun_out/r5s2/shard.txt
timeout 300 bash tools/shard_timeline.sh > gpurun_out/r5s2/timeline_default.txt 2>&1
cp gpurun_out/stl/*kernel_trace.csv gpurun_out/r5s2/tl_default_kernels.csv; cp gpurun_out/stl/*memory_copy_trace.csv gpurun_out/r5s2/tl_default_copies.csv
timeout 300 bash tools/shard_timeline.sh pinned=1 > gpurun_out/r5s2/timeline_pinned.txt 2>&1
cat gpurun_out/r5s2/timeline_default.txt gpurun_out/r5s2/timeline_pinned.txt
JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so timeout 120 python tools/shard_trace.py 128 > gpurun_out/r5s2/trace_default.txt 2>&1
tail -40 gpurun_out/r5s2/trace_default.txt
