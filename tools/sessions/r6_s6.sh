#!/bin/bash
# round 6, session 6: device-side timelines of the shard with the FIFO as it is now (1 staging thread, 1 copy stream)
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_s6; mkdir -p $O
export JGA_LIB_PATH=$PWD/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for cfg in "2 1" "1 0" "0 1"; do
  set -- $cfg
  echo "=========== JGA_PIPE_SHORT_FIFO=$1 pinned=$2" >> $O/timelines.txt
  JGA_PIPE_SHORT_FIFO=$1 bash tools/shard_timeline.sh pinned=$2 >> $O/timelines.txt 2>&1
  echo "--- host trace" >> $O/timelines.txt
  JGA_PIPE_SHORT_FIFO=$1 timeout 200 python tools/shard_trace.py 128 pinned=$2 2>&1 | grep "TOTAL\|heads + blob\|lane group" | tail -28 >> $O/timelines.txt
done
grep -n "=====\|LAST RUN\|link busy\|^  [0-9]\|TOTAL\|heads + blob\|span\|first start\|hj_unstuff_count\|jga_idct" $O/timelines.txt | cut -c1-230
