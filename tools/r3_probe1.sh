#!/bin/bash
# round 3, first GPU call: suite at HEAD, per-kernel baseline of the entropy stage, and the timing
# probe "synchronisation runs without the DC sums" (variants/nodc.so: wrong planes, timing only)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3_probe1; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
for rep in 1 2 3; do for v in base nodc; do
  echo -n "$v "; JGA_LIB_PATH=$PWD/jpeg_gpu_amd/variants/$v.so timeout 200 python tools/hbench.py 2>&1 | grep "x48" | tail -1
done; done | tee $OUT/hbench.txt
for v in base nodc; do
  rm -rf $OUT/prof_$v
  JGA_LIB_PATH=$PWD/jpeg_gpu_amd/variants/$v.so timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$v -o h -f csv -- python tools/hbench.py > $OUT/prof_$v.txt 2>&1
  python3 - <<PY | tee -a $OUT/kernels.txt
import csv,glob
fn=glob.glob("$OUT/prof_$v/**/h_kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(fn)))
print("== $v")
for name in ("hj_sync_round","hj_sync_sparse","hj_write","hj_scan","hj_init","jga_idct","fillBuffer"):
    r=[x for x in rows if name in x["Kernel_Name"]]
    print(name,[round((int(x["End_Timestamp"])-int(x["Start_Timestamp"]))/1e3) for x in r][-12:])
PY
  rm -rf $OUT/prof_$v
done
