#!/bin/bash
# sweep the write-pass flush threshold: prints hj_write time per setting
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for f in ${@:-1 4 8 12 16 24 32}; do
  echo "== JGA_HUFF_FLUSH=$f"
  JGA_HUFF_FLUSH=$f ./tools/hprof.sh | grep -E "hj_write|huffman|equal" | tail -3
done
