#!/bin/bash
# the list rounds forced onto every batch (tuning build, JGA_HUFF_LIST=1): corrupted scans and random batches against the host stage
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
echo "## JGA_HUFF_LIST=1 tools/fuzz_gpu_huff.py 121 ${1:-3000} ; 122 ${2:-1500} wide"
env JGA_LIB_PATH=$T JGA_HUFF_LIST=1 python tools/fuzz_gpu_huff.py 121 ${1:-3000} 2>&1 | tail -2
env JGA_LIB_PATH=$T JGA_HUFF_LIST=1 python tools/fuzz_gpu_huff.py 122 ${2:-1500} wide 2>&1 | tail -2
echo "## JGA_HUFF_LIST=1,3 tools/soak_gpu_huff.py ${3:-200} 17   (three steps inside a workgroup: more launches per decode)"
env JGA_LIB_PATH=$T JGA_HUFF_LIST=1,3 python tools/soak_gpu_huff.py ${3:-200} 17 2>&1 | tail -3
echo "## JGA_HUFF_LIST=1 tools/periodic_streams.py"
env JGA_LIB_PATH=$T JGA_HUFF_LIST=1 python tools/periodic_streams.py 2>&1 | tail -8
