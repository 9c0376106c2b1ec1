#!/bin/bash
# round 6, session 1: the pipeline's tests after the input-cache / FIFO-upload changes, photo-like parity, and the
# 128-file shard with the short runs' uploads FIFO on two shared copy streams against turns on the link
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_s1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py tests/test_harness.py -x -q -m gpu -k "pipeline or input_cache or photo or config4 or heap" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
export JGA_LIB_PATH=$PWD/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for rep in 1 2; do
  echo "== fifo (default) $rep" >> $O/shard.txt
  timeout 300 python tools/shard_sweep.py 128 "" >> $O/shard.txt 2>&1
  echo "== turns (JGA_PIPE_SHORT_FIFO=0) $rep" >> $O/shard.txt
  JGA_PIPE_SHORT_FIFO=0 timeout 300 python tools/shard_sweep.py 128 "" >> $O/shard.txt 2>&1
done
for n in 32 64 256; do
  echo "== fifo n=$n" >> $O/shard.txt
  timeout 300 python tools/shard_sweep.py $n "" >> $O/shard.txt 2>&1
  echo "== turns n=$n" >> $O/shard.txt
  JGA_PIPE_SHORT_FIFO=0 timeout 300 python tools/shard_sweep.py $n "" >> $O/shard.txt 2>&1
done
cat $O/shard.txt
