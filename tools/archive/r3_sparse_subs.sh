#!/bin/bash
# (needs the templated build of hj_sync_sparse that was measured and not kept: profiles/r3_entropy_stage_steps.md)
# subsequences per wave of the sparse launches
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
JGA_HUFF_SPARSE_SUBS=512,1024,1024 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "huff or split or corrupt or periodic or irregular" 2>&1 | tail -2
for rep in 1 2; do for cfg in 256,256,256 256,512,1024 512,1024,1024 512,512,512 1024,1024,1024; do
  echo -n "$cfg x48 :: "; JGA_HUFF_SPARSE_SUBS=$cfg timeout 120 python tools/hbench.py 2>&1 | grep "Mpix/s" | tail -1 | sed 's/.*| huffman/huffman/'
  echo -n "$cfg x16 :: "; JGA_HUFF_SPARSE_SUBS=$cfg timeout 120 python tools/hbench.py 3840 2160 420 16 2>&1 | grep "Mpix/s" | tail -1 | sed 's/.*| huffman/huffman/'
  echo -n "$cfg light x32 :: "; CONTENT=light JGA_HUFF_SPARSE_SUBS=$cfg timeout 120 python tools/hbench.py 3840 2160 420 32 2>&1 | grep "Mpix/s" | tail -1 | sed 's/.*| huffman/huffman/'
done; done
for cfg in 256,256,256 512,1024,1024; do JGA_HUFF_SPARSE_SUBS=$cfg bash tools/hprof.sh 2>&1 | grep "hj_sync_sparse"; done
