#!/bin/bash
# round 3: PMC of the entropy-stage kernels (one counter group per pass), per-wave figures
# usage: tools/r3_pmc.sh TAG [hbench args]      (env for the run is inherited)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $OUT; shift
CMD="python tools/hbench.py ${@:-3840 2160 420 48 0}"
[ -f gpurun_out/counters_avail.txt ] || (timeout 100 rocprofv3 -L > gpurun_out/counters_avail.txt 2>&1)
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT -o a -f csv -- $CMD > /dev/null 2>$OUT/a.err
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH -d $OUT -o b -f csv -- $CMD > /dev/null 2>$OUT/b.err
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE -d $OUT -o c -f csv -- $CMD > /dev/null 2>$OUT/c.err
timeout 200 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE -d $OUT -o d -f csv -- $CMD > /dev/null 2>$OUT/d.err
tail -2 $OUT/c.err $OUT/d.err
python3 - <<PY
import csv,collections,glob
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("$OUT/*_counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"].split("(")[0]
        if "hj_" not in k: continue
        agg[k][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
for k,c in sorted(agg.items()):
    # the longest launch of each kernel
    pick={n: max(v,key=lambda x:x[1]) for n,v in c.items()}
    g=lambda n: pick[n][0] if n in pick else None
    dur=max(d for _,d in pick.values())/1e3
    print("%s  (longest launch %.0f us)" % (k, dur))
    w=g("SQ_WAVES")
    if w:
        print("   waves %.0f | per wave: VALU %.0f SALU %.0f LDS %.0f branch %s vmem_rd %s vmem_wr %s" % (w, g("SQ_INSTS_VALU")/w, g("SQ_INSTS_SALU")/w, g("SQ_INSTS_LDS")/w,
              "%.0f"%(g("SQ_INSTS_BRANCH")/w) if g("SQ_INSTS_BRANCH") else "-", "%.0f"%(g("SQ_INSTS_VMEM_RD")/w) if g("SQ_INSTS_VMEM_RD") else "-", "%.0f"%(g("SQ_INSTS_VMEM_WR")/w) if g("SQ_INSTS_VMEM_WR") else "-"))
    gui=g("GRBM_GUI_ACTIVE")
    if gui:
        cyc=gui/8
        for n in ("SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_ACTIVE_INST_SCA","SQ_ACTIVE_INST_VMEM","SQ_ACTIVE_INST_MISC","SQ_ACTIVE_INST_ANY","SQ_LDS_BANK_CONFLICT","SQ_LDS_ADDR_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_WAIT_INST_LDS","SQ_WAIT_INST_ANY","SQ_WAIT_ANY","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_INST_LEVEL_VMEM","SQ_INST_LEVEL_LDS"):
            if g(n) is not None: print("   %-24s %.4g   /(1024 SIMDs x cycles) = %.3f   x4 = %.3f   /(256 CUs x cycles) = %.3f" % (n, g(n), g(n)/(1024*cyc), 4*g(n)/(1024*cyc), g(n)/(256*cyc)))
        for n in ("TA_BUSY_avr","TA_TA_BUSY_sum","TCP_TOTAL_CACHE_ACCESSES_sum","TCP_TCC_READ_REQ_sum","TA_FLAT_READ_WAVEFRONTS_sum","TA_ADDR_STALLED_BY_TC_CYCLES_sum","TCP_PENDING_STALL_CYCLES_sum"):
            if g(n) is not None: print("   %-24s %.4g   /cycles = %.3f  /(256 x cycles) = %.3f" % (n, g(n), g(n)/cyc, g(n)/(256*cyc)))
        print("   cycles per XCD %.4g (%.2f GHz)" % (cyc, cyc/dur/1e3))
PY
