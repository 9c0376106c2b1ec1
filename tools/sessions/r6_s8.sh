#!/bin/bash
# round 6, session 8: fuzz / soak / leak after the entropy stage's pruning and policy changes and the pipeline's
# registration changes (-> profiles/r6_fuzz_soak_leak.txt)
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_s8; mkdir -p $O; R=$O/fuzz_soak_leak.txt
t() { echo "## $*" >> $R; ( time timeout 900 "$@" ) >> $R 2>&1; }
t python tools/fuzz_gpu_huff.py 91 4000
t python tools/fuzz_gpu_huff.py 92 3000 wide
t python tools/soak_gpu_huff.py 400 21
t python tools/periodic_streams.py
t python tools/soak_pipeline.py 80 11
t python tools/soak_input_cache.py 30 6
t python tools/leak_check.py 10
export JGA_LIB_PATH=$PWD/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
echo "## the same with list rounds forced onto every batch (JGA_HUFF_LIST=1, tuning build)" >> $R
JGA_HUFF_LIST=1 t python tools/fuzz_gpu_huff.py 93 3000
JGA_HUFF_LIST=1 t python tools/soak_gpu_huff.py 200 22
JGA_HUFF_LIST=1 t python tools/periodic_streams.py
grep -v "amdgpu.ids" $R | cut -c1-400 | tail -80
