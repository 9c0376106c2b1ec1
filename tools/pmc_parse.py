"""Summarise rocprofv3 --pmc CSVs (one or more *_counter_collection.csv) per kernel."""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for fn in sorted(glob.glob(d + "/*_counter_collection.csv")):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        m = re.search(r"jga_idct_(\w+)_kernel<([^>]*)>", k)
        if not m:
            continue
        key = m.group(1) + "<" + m.group(2).replace(" ", "") + ">"
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for key in sorted(agg):
    c = {n: sum(v) / len(v) for n, v in agg[key].items()}
    t = sum(dur[key]) / len(dur[key])
    print("%s  dur %.1f us" % (key, t / 1e3))
    g = c.get
    if g("SQ_WAVES"):
        w = g("SQ_WAVES")
        print("   waves %.0f valu/wave %.0f  wavecyc/wave %.0f  VALUactive/wavecyc %.2f  wait_any %.2f wait_inst %.2f active_any %.2f"
              % (w, g("SQ_INSTS_VALU") / w, g("SQ_WAVE_CYCLES") / w,
                 g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
                 g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")))
        if g("GRBM_GUI_ACTIVE"):
            gui = g("GRBM_GUI_ACTIVE")
            print("   GRBM_GUI_ACTIVE %.3g (=> %.2f GHz)  VALU busy per SIMD = ACTIVE_VALU*4/(1024*GUI) = %.2f   BUSY_CYCLES %.3g"
                  % (gui, gui / t, g("SQ_ACTIVE_INST_VALU") * 4 / (1024 * gui), g("SQ_BUSY_CYCLES")))
    for n in ("SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR", "SQ_INST_CYCLES_VMEM_RD",
              "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS",
              "FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum",
              "TCC_EA0_WRREQ_64B_sum", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INST_LEVEL_VMEM",
              "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TCR_TCP_STALL_CYCLES_sum", "TCC_EA0_WRREQ_STALL_sum",
              "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum",
              "TCP_TOTAL_CACHE_ACCESSES_sum"):
        if g(n) is not None:
            print("   %-26s %.4g" % (n, g(n)))
