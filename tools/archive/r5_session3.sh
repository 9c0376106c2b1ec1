#!/bin/bash
# round 5, session 3: does a kernel fetching host memory slow its neighbours?  the shard by fetch grid
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s3
timeout 300 tools/bin/r5_probes contend > gpurun_out/r5s3/contend.txt 2>&1
cat gpurun_out/r5s3/contend.txt
export JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for w in 8 16 32 64 160; do
  echo "== JGA_FETCH_WGS=$w" >> gpurun_out/r5s3/shard.txt
  JGA_FETCH_WGS=$w timeout 300 python tools/shard_sweep.py 128 "" >> gpurun_out/r5s3/shard.txt 2>&1
done
echo "== JGA_PIPE_FETCH=0 (copy calls, one wait per group)" >> gpurun_out/r5s3/shard.txt
JGA_PIPE_FETCH=0 timeout 300 python tools/shard_sweep.py 128 "" "unstuff=1" >> gpurun_out/r5s3/shard.txt 2>&1
for q in 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q JGA_FETCH_WGS=16" >> gpurun_out/r5s3/shard.txt
  GPU_MAX_HW_QUEUES=$q JGA_FETCH_WGS=16 timeout 300 python tools/shard_sweep.py 128 "" "unstuff=1" >> gpurun_out/r5s3/shard.txt 2>&1
done
cat gpurun_out/r5s3/shard.txt
JGA_FETCH_WGS=16 timeout 300 bash tools/shard_timeline.sh > gpurun_out/r5s3/timeline_w16.txt 2>&1
cp gpurun_out/stl/*kernel_trace.csv gpurun_out/r5s3/tl_w16_kernels.csv; cp gpurun_out/stl/*memory_copy_trace.csv gpurun_out/r5s3/tl_w16_copies.csv
cat gpurun_out/r5s3/timeline_w16.txt
