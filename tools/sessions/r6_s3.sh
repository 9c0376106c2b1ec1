#!/bin/bash
# round 6, session 3: (1) the 128-file shard under the FIFO knobs; (2) hj_write with two AC symbols per look-up
# (variants/pairs.so) against the tree's, recipe and photograph-like content, with per-kernel times
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_s3; mkdir -p $O
export JGA_LIB_PATH=$PWD/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
run() { echo "== $*" >> $O/shard.txt; env "$@" timeout 300 python tools/shard_sweep.py 128 "" >> $O/shard.txt 2>&1; }
for rep in 1 2; do
  run JGA_PIPE_SHORT_FIFO=0
  run JGA_PIPE_SHORT_FIFO=0 JGA_PIPE_SHORT_RAMP=1
  run JGA_PIPE_SHORT_FIFO=1
  run JGA_PIPE_SHORT_FIFO=1 JGA_PIPE_SHORT_RAMP=1
  run JGA_PIPE_SHORT_FIFO=1 JGA_PIPE_SHORT_RAMP=1 JGA_PIPE_FIFO_STREAMS=2
  run JGA_PIPE_SHORT_FIFO=1 JGA_PIPE_SHORT_RAMP=1 JGA_PIPE_FIFO_THREADS=3
done
cat $O/shard.txt
# (2)
for content in recipe photo; do
 for cfg in "3840 2160 420 48 0" "1920 1080 420 128 0"; do
  for pass in 1 2 3; do
    for f in jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so jpeg_gpu_amd/variants/pairs.so; do
      echo "== $content $cfg | $(basename $f)" >> $O/pairs.txt
      env CONTENT=$content JGA_LIB_PATH=$PWD/$f timeout 300 python tools/hbench.py $cfg 2>&1 | grep -E "huffman|equal" | tail -3 >> $O/pairs.txt
    done
  done
 done
done
cat $O/pairs.txt
for f in jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so jpeg_gpu_amd/variants/pairs.so; do
  for content in recipe photo; do
    echo "== kernels: $content | $(basename $f)" >> $O/pairs_kernels.txt
    env CONTENT=$content JGA_LIB_PATH=$PWD/$f bash tools/hprof.sh 3840 2160 420 48 0 >> $O/pairs_kernels.txt 2>&1
  done
done
cat $O/pairs_kernels.txt | cut -c1-300
