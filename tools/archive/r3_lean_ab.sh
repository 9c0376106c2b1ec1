#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for lean in 0 1; do
  echo -n "lean=$lean x48 :: "; JGA_HUFF_LEAN=$lean timeout 120 python tools/hbench.py 2>&1 | grep "Mpix/s" | tail -1 | sed 's/.*| huffman/huffman/'
  echo -n "lean=$lean x8 :: "; JGA_HUFF_LEAN=$lean timeout 120 python tools/hbench.py 3840 2160 420 8 2>&1 | grep "Mpix/s" | tail -1 | sed 's/.*| huffman/huffman/'
  echo -n "lean=$lean light x32 :: "; CONTENT=light JGA_HUFF_LEAN=$lean timeout 120 python tools/hbench.py 3840 2160 420 32 2>&1 | grep "Mpix/s" | tail -1 | sed 's/.*| huffman/huffman/'
done; done
for lean in 0 1; do JGA_HUFF_LEAN=$lean bash tools/hprof.sh 2>&1 | grep "hj_sync_round\|hj_write"; done
