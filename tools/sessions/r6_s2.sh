#!/bin/bash
# round 6, session 2: device-side timelines of the 128-file shard: FIFO uploads against turns, pinned and pageable files
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_s2; mkdir -p $O
export JGA_LIB_PATH=$PWD/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for mode in "fifo 1" "turns 0"; do
  set -- $mode
  for pin in 1 0; do
    echo "=========== $1 pinned=$pin" >> $O/timelines.txt
    JGA_PIPE_SHORT_FIFO=$2 bash tools/shard_timeline.sh pinned=$pin >> $O/timelines.txt 2>&1
    echo "--- host trace" >> $O/timelines.txt
    JGA_PIPE_SHORT_FIFO=$2 timeout 200 python tools/shard_trace.py 128 pinned=$pin 2>&1 | tail -45 >> $O/timelines.txt
  done
done
cat $O/timelines.txt
