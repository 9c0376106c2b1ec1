#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for it in ${@:-"3,2,6" "4,3,6" "4,4,6" "5,3,6" "6,4,6" "8,8,6" "2,2,8" "3,3,6"}; do
  echo "== JGA_HUFF_ITERS=$it"
  JGA_HUFF_ITERS=$it ./tools/hprof.sh | grep -E "hj_sync|huffman" | tail -2 | cut -c1-150
done
