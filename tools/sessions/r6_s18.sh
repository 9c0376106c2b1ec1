#!/bin/bash
# round 6, session 18: hj_write with the next symbol's table in a register and the DC difference read back from the
# block (HJ_WRITE_LEAN, the tree) against the loop as it was (variants/writeold.so)
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_s18; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_list_rounds.py tests/test_rare_sampling.py -x -q -m gpu -k "huff or gpu_entropy or list or rare or dense or periodic or damaged" 2>&1 | tail -3
for content in recipe photo; do
  for pass in 1 2 3; do
    for f in jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so jpeg_gpu_amd/variants/writeold.so; do
      echo "== $content | $(basename $f)" >> $O/ab.txt
      env CONTENT=$content JGA_LIB_PATH=$PWD/$f timeout 300 python tools/hbench.py 3840 2160 420 48 0 2>&1 | grep -E "huffman|equal" | tail -3 >> $O/ab.txt
    done
  done
done
grep -v equal $O/ab.txt | cut -c1-160
for f in jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so jpeg_gpu_amd/variants/writeold.so; do
  echo "== kernels | $(basename $f)"; env JGA_LIB_PATH=$PWD/$f bash tools/hprof.sh 3840 2160 420 48 0 2>&1 | grep "hj_write \|hj_sync_round"
done
timeout 300 python tools/fuzz_gpu_huff.py 95 2000 2>&1 | tail -1
