#!/bin/bash
# lone frames against the round knobs: which kernel runs the later rounds, in-group iterations
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "" "JGA_HUFF_SPARSE_FROM=99" "JGA_HUFF_SPARSE_FROM=99 JGA_HUFF_ITERS=6,6,6" "JGA_HUFF_ITERS=6,6,6" "JGA_HUFF_ITERS=4,2,6" "JGA_HUFF_SPARSE_FROM=2" "JGA_HUFF_SPARSE_FROM=2 JGA_HUFF_ITERS=4,4,6" "JGA_HUFF_SPARSE_FROM=99 JGA_HUFF_ITERS=10,10,6"; do
  for geo in "1920 1080 420 1" "3840 2160 420 1" "3840 2160 444 1"; do
    echo -n "[$cfg] $geo :: "; env $cfg timeout 120 python tools/hbench.py $geo 2>&1 | grep "Mpix/s" | tail -1 | sed 's/.*| huffman/huffman/'
  done
done
