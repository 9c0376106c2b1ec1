#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in 1024 -1 0; do
  echo "== scale-proxy child, --input-cache-MB $c"; timeout 400 python bench.py --as-rank-of 8 --input-cache-MB $c 2>/dev/null | grep SCALE_PROXY | python -c "
import sys, json
d = json.loads(sys.stdin.read()[len('SCALE_PROXY '):]); print({k: d[k] for k in ('pageable', 'pinned')})"
done > gpurun_out/r4s11_direct.txt 2>&1
(timeout 600 python -m pytest tests/test_harness.py -m gpu -q -x 2>&1 | tail -3) >> gpurun_out/r4s11_direct.txt
cat gpurun_out/r4s11_direct.txt
