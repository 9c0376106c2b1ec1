#!/bin/bash
# round 3: interleaved hbench over the library builds in jpeg_gpu_amd/variants/*.so + per-kernel trace of each
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r3var}; OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ -n "$RUN_TESTS" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q -k "${KEXPR:-huffman or pipeline or config or unstuff or harness or irregular or extreme}" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
fi
for rep in 1 2 3; do for f in jpeg_gpu_amd/variants/*.so; do
  echo -n "$(basename $f) "; JGA_LIB_PATH=$PWD/$f timeout 200 python tools/hbench.py ${HB_ARGS} 2>&1 | grep "Mpix/s" | tail -1
done; done | tee $OUT/hbench.txt
for f in jpeg_gpu_amd/variants/*.so; do
  v=$(basename $f .so); rm -rf $OUT/prof_$v
  JGA_LIB_PATH=$PWD/$f timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$v -o h -f csv -- python tools/hbench.py ${HB_ARGS} > $OUT/prof_$v.txt 2>&1
  python3 - <<PY | tee -a $OUT/kernels.txt
import csv,glob
fn=glob.glob("$OUT/prof_$v/**/h_kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(fn)))
print("== $v")
for name in ("hj_sync_round","hj_sync_sparse","hj_write","hj_scan","hj_init","hj_dc","jga_idct","fillBuffer"):
    r=[x for x in rows if name in x["Kernel_Name"]]
    if r: print(name,[round((int(x["End_Timestamp"])-int(x["Start_Timestamp"]))/1e3) for x in r][-12:])
PY
  rm -rf $OUT/prof_$v
done
