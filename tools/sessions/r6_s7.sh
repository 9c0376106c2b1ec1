#!/bin/bash
# round 6, session 7: the shard with every group through the FIFO (no named first groups), one copy call per group
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_s7; mkdir -p $O
export JGA_LIB_PATH=$PWD/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
run() { echo "== $*" >> $O/shard.txt; env "$@" timeout 300 python tools/shard_sweep.py 128 "" >> $O/shard.txt 2>&1; }
for rep in 1 2 3; do
  run JGA_PIPE_SHORT_FIFO=0
  run JGA_PIPE_SHORT_FIFO=2
  run JGA_PIPE_SHORT_FIFO=2 JGA_PIPE_SHORT_RAMP=1
  run JGA_PIPE_SHORT_FIFO=2 JGA_PIPE_FIFO_THREADS=2
  run JGA_PIPE_SHORT_FIFO=2 JGA_PIPE_SHORT_RAMP=1 JGA_PIPE_FIFO_THREADS=2
  run JGA_PIPE_SHORT_FIFO=2 JGA_PIPE_FIFO_NAMED=2
done
cat $O/shard.txt
echo "=========== timeline FIFO=2 RAMP=1 pinned=1" >> $O/timelines.txt
JGA_PIPE_SHORT_FIFO=2 JGA_PIPE_SHORT_RAMP=1 bash tools/shard_timeline.sh pinned=1 >> $O/timelines.txt 2>&1
echo "=========== timeline FIFO=2 pinned=0" >> $O/timelines.txt
JGA_PIPE_SHORT_FIFO=2 bash tools/shard_timeline.sh pinned=0 >> $O/timelines.txt 2>&1
grep -n "=====\|LAST RUN\|link busy\|^  [0-9]\|span" $O/timelines.txt | cut -c1-230
