#!/bin/bash
# PMC of one kernel: tools/pmc_kernel.sh <kernel-name-substring> <outdir> -- <command...>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
K=$1; OUT=gpurun_out/$2; shift 3; mkdir -p $OUT
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT -o a -f csv -- "$@" > /dev/null 2>$OUT/a.err
timeout 150 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT -o b -f csv -- "$@" > /dev/null 2>$OUT/b.err
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -d $OUT -o c -f csv -- "$@" > /dev/null 2>$OUT/c.err
python3 - <<PY
import csv,collections,glob
agg=collections.defaultdict(list); dur=[]
for fn in glob.glob("$OUT/*_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        if "$K" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
print("$K: %d samples, avg %.1f us" % (len(dur), sum(dur)/max(1,len(dur))/1e3))
c={k:sum(v)/len(v) for k,v in agg.items()}
for k in sorted(c): print("  %-24s %.5g" % (k, c[k]))
g=c.get("GRBM_GUI_ACTIVE",0)/8
if g:
    print("  VALU busy %.2f  LDS busy %.2f  VMEM busy %.2f" % (c.get("SQ_ACTIVE_INST_VALU",0)*4/(1024*g), c.get("SQ_ACTIVE_INST_LDS",0)*4/(256*g), c.get("SQ_ACTIVE_INST_VMEM",0)*4/(256*g)))
if "FETCH_SIZE" in c: print("  HBM read %.4g B (2*FETCH_SIZE*1024), write %.4g B" % (c["FETCH_SIZE"]*2048, c.get("WRITE_SIZE",0)*1024))
PY
