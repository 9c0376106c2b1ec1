#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $OUT; shift
CMD="python tools/hbench.py ${@:-3840 2160 420 48 0}"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT -o a -f csv -- $CMD > /dev/null 2>$OUT/a.err
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH -d $OUT -o b -f csv -- $CMD > /dev/null 2>$OUT/b.err
python3 - <<PY
import csv,collections,glob
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("$OUT/*_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"].split("(")[0]
        if "hj_" not in k: continue
        # only the big first round for sync
        agg[k][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
for k,c in agg.items():
    print(k)
    for n,v in sorted(c.items()):
        big=sorted(v,key=lambda x:-x[1])[:3]
        print("   %-22s max-dur launches: %s" % (n, ["%.4g (%.0fus)"%(a,d/1e3) for a,d in big]))
PY
