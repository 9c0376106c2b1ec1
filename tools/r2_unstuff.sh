#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r2unstuff}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "unstuff or corrupted or pipeline or config4 or irregular" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.log
for thr in 2 4 24; do for v in "1 0" "2 0" "0 1"; do set -- $v
  echo -n "unstuff=$1 pinned=$2 threads=$thr: "; UNSTUFF=$1 PINNED=$2 SWEEP_CFGS="32,8,$thr" timeout 200 python tools/e2e_sweep2.py 2304; done; done | tee $OUT/e2e.txt
