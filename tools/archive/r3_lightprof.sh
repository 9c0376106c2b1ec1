cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3_light; mkdir -p $OUT
CONTENT=light timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o h -f csv -- python tools/hbench.py 3840 2160 420 32 0 > $OUT/hbench.txt 2>&1
grep "Mpix/s\|equal" $OUT/hbench.txt | tail -3
python3 - <<PY
import csv,glob
fn=glob.glob("$OUT/prof/**/h_kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(fn)))
for name in ("hj_init","hj_sync_round","hj_sync_sparse","hj_scan","hj_write","hj_dc_scan","hj_dc_apply","jga_idct","fillBuffer"):
    r=[x for x in rows if name in x["Kernel_Name"]]
    if r: print(name,[round((int(x["End_Timestamp"])-int(x["Start_Timestamp"]))/1e3) for x in r][-14:])
PY
rm -rf $OUT/prof
