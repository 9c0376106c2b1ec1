#!/bin/bash
# round 3: full GPU suite, hbench x3, per-kernel trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r3_full}; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q ${KEXPR:+-k "$KEXPR"} > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
for rep in 1 2 3; do timeout 200 python tools/hbench.py ${HB_ARGS} 2>&1 | grep "Mpix/s\|equal" | tail -3; done | tee $OUT/hbench.txt
rm -rf $OUT/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o h -f csv -- python tools/hbench.py ${HB_ARGS} > $OUT/prof.txt 2>&1
python3 - <<PY | tee $OUT/kernels.txt
import csv,glob
fn=glob.glob("$OUT/prof/**/h_kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(fn)))
for name in ("hj_sync_round","hj_sync_sparse","hj_write","hj_scan","hj_init","hj_dc_scan","hj_dc_apply","jga_idct","fillBuffer"):
    r=[x for x in rows if name in x["Kernel_Name"]]
    if r: print(name,[round((int(x["End_Timestamp"])-int(x["Start_Timestamp"]))/1e3) for x in r][-14:])
PY
rm -rf $OUT/prof
