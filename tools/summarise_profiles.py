"""Turn gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the committed
profiles/<tag>_* summaries and profiles/pmc_latest.json (read by bench.py for
roofline.traffic)."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = os.path.join("gpurun_out", "prof_" + tag)
dst = "profiles"
os.makedirs(dst, exist_ok=True)
KERNEL = "jga_idct_rgb_kernel"


def rows(name):
    p = os.path.join(src, name)
    return list(csv.DictReader(open(p))) if os.path.exists(p) else []


bench = json.load(open(os.path.join(src, "bench.json")))
batch = bench["config"]["batch_per_gpu"]
out = ["# rocprofv3 summary, tag %s — command: rocprofv3 --kernel-trace --stats -- "
       "python bench.py --batch %d --steps 20 --warmup 3 --no-cpu --no-e2e" % (tag, batch), ""]
st = rows("stats_kernel_stats.csv")
out.append("## kernel stats (rocprofv3 --kernel-trace --stats)")
out.append("| kernel | calls | total ns | avg ns | min ns | max ns | % |")
out.append("|---|---|---|---|---|---|---|")
avg_ns = None
best_calls = 0
for r in st:
    out.append("| %s | %s | %s | %s | %s | %s | %s |" % (
        r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"],
        r["MaxNs"], r["Percentage"]))
    if KERNEL in r["Name"] and int(r["Calls"]) > best_calls:   # the bench kernel: most calls
        best_calls = int(r["Calls"])
        avg_ns = float(r["AverageNs"])
out.append("")
pm = collections.defaultdict(list)
for f in ("fetch", "write", "tcc", "sq"):
    for r in rows(f + "_counter_collection.csv"):
        if KERNEL in r["Kernel_Name"]:
            pm[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in pm.items()}
out.append("## PMC, per launch of %s (averages over the timed launches)" % KERNEL)
for k in sorted(c):
    out.append("- %s = %.6g" % (k, c[k]))
traffic = None
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB-like units of 1024 B;
    # on gfx950 FETCH_SIZE reports exactly half of a wide coalesced read stream -> x2.
    fetch = c["FETCH_SIZE"] * 1024 * 2
    write = c["WRITE_SIZE"] * 1024
    traffic = fetch + write
    alg = bench["roofline"]["algorithmic_bytes_per_launch"]
    out.append("")
    out.append("HBM traffic per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 = %.4g + %.4g = %.4g B "
               "(algorithmic %.4g B, ratio %.3f)" % (fetch, write, traffic, alg, traffic / alg))
    if "TCC_EA0_RDREQ_sum" in c:
        out.append("cross-check: TCC_EA0_RDREQ*128 B = %.4g B, TCC_EA0_WRREQ_64B*64 B = %.4g B"
                   % (c["TCC_EA0_RDREQ_sum"] * 128, c.get("TCC_EA0_WRREQ_64B_sum", 0) * 64))
if avg_ns:
    out.append("")
    out.append("kernel average duration (rocprof) %.1f us vs bench.py HIP-event %.1f us"
               % (avg_ns / 1e3, bench["roofline"]["kernel_ms_per_launch"] * 1e3))
open(os.path.join(dst, "%s_rocprof_summary.md" % tag), "w").write("\n".join(out) + "\n")
json.dump(bench, open(os.path.join(dst, "%s_bench.json" % tag), "w"), indent=1)
if traffic:
    latest = {"workload": "3840x2160 420", "batch": batch, "hbm_bytes_per_launch": int(traffic),
              "fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"],
              "source": "profiles/%s_rocprof_summary.md" % tag}
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c and "SQ_WAVES" in c:
        # vector-ALU side of the same launch: wave instructions, and the busy fraction by the
        # 4-cycles-per-wave64-instruction convention (1024 SIMDs; GRBM_GUI_ACTIVE sums 8 XCDs)
        latest["valu_insts_per_launch"] = int(c["SQ_INSTS_VALU"])
        latest["valu_insts_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
        latest["valu_busy_4clk"] = round(c.get("SQ_ACTIVE_INST_VALU", c["SQ_INSTS_VALU"]) * 4
                                         / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 3)
        out.append("VALU: %d wave-instructions per launch (%.0f per wave); busy %.2f of the SIMD cycles at "
                   "4 clk per instruction" % (latest["valu_insts_per_launch"], latest["valu_insts_per_wave"],
                                             latest["valu_busy_4clk"]))
    json.dump(latest, open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
open(os.path.join(dst, "%s_rocprof_summary.md" % tag), "w").write("\n".join(out) + "\n")
print("\n".join(out))
