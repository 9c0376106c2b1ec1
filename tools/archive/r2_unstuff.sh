#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2unstuff}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "unstuff or corrupted or pipeline or config4 or irregular" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for rep in 1 2; do for v in "1 0" "2 0" "0 1"; do set -- $v
  echo -n "unstuff=$1 pinned=$2: "; UNSTUFF=$1 PINNED=$2 SWEEP_CFGS="32,8,24" timeout 200 python tools/e2e_sweep2.py 2304; done; done | tee $OUT/e2e.txt
timeout 400 python bench.py --no-cpu --no-pack --no-other --no-gpu-entropy 2>$OUT/bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench value', d['value'], 'pageable', d['e2e']['pageable_files_to_rgb_hbm']['value'], 'north star', d['e2e']['north_star_host_huffman_to_rgb_hbm']['value'])"
