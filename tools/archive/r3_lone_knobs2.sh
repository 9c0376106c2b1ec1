#!/bin/bash
# where the sparse kernel starts to pay: batch size against the kernel of the later rounds
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for geo in "3840 2160 420 2" "3840 2160 420 4" "3840 2160 420 8" "3840 2160 420 16" "1920 1080 420 8" "1920 1080 420 32" "3840 2160 444 4" "7680 4320 420 1 -1"; do
  for cfg in "JGA_HUFF_SPARSE_FROM=1" "JGA_HUFF_SPARSE_FROM=99" "JGA_HUFF_SPARSE_FROM=2 JGA_HUFF_ITERS=4,4,6"; do
    echo -n "$geo [$cfg] :: "; env $cfg timeout 120 python tools/hbench.py $geo 2>&1 | grep "Mpix/s" | tail -1 | sed 's/.*| huffman/huffman/'
  done
done
