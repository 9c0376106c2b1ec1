#!/usr/bin/env python3
"""HBM traffic (and the VALU side) of the fused IDCT+RGB kernel from rocprofv3 PMC passes.

    python tools/pmc_traffic.py [--batch 48] [--out gpurun_out/pmc_traffic]

Runs `tools/kbench.py --child 3840 2160 420 <batch>` (the kernel on a batch of coefficient planes
resident in HBM: 2 x 20 launches) under rocprofv3, one counter group per pass as
MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc only
with --kernel-trace), applies its gfx950 corrections (FETCH_SIZE / WRITE_SIZE are in units of
1024 B; FETCH_SIZE reports half of a wide coalesced read stream -> x2), writes
profiles/pmc_latest.json (read by bench.py for roofline.traffic, with the HEAD and date it was
taken at) and prints `PMC <json>`.  bench.py --measure-traffic calls this."""
import argparse
import collections
import csv
import datetime
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "jga_idct_rgb_kernel"
PASSES = {
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "tcc": ["TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"],
    "sq": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=48)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pmc_traffic"))
    args = ap.parse_args()
    args.out = os.path.abspath(args.out)         # (rocprofv3 runs with /tmp as its directory)
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        sys.exit("rocprofv3 not found")
    os.makedirs(args.out, exist_ok=True)
    child = [sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "--child",
             "3840", "2160", "420", str(args.batch)]
    env = dict(os.environ, TMPDIR="/tmp")
    c = {}
    durs = []
    for name, counters in PASSES.items():
        cmd = [prof, "--kernel-trace", "--pmc"] + counters + ["-d", args.out, "-o", name, "-f", "csv",
                                                              "--"] + child
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=240)
        if r.returncode:
            sys.stderr.write(r.stderr[-2000:])
            sys.exit("rocprofv3 pass '%s' failed" % name)
        vals = collections.defaultdict(list)
        for fn in glob.glob(os.path.join(args.out, "**", "*" + name + "_counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(fn)):
                if KERNEL in row["Kernel_Name"]:
                    vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
                    if name == "fetch":
                        durs.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        for k, v in vals.items():
            c[k] = sum(v) / len(v)
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        sys.exit("no %s launches in the counter files" % KERNEL)
    traffic = c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE,
                          stderr=subprocess.DEVNULL, text=True).stdout.strip() or None
    latest = {"workload": "3840x2160 420", "batch": args.batch, "hbm_bytes_per_launch": int(traffic),
              "fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"],
              "source": "tools/pmc_traffic.py (rocprofv3 --kernel-trace --pmc, one group per pass)",
              "head": head or os.environ.get("JGA_HEAD"),
              "date": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"),
              "kernel_avg_us_under_pmc": round(sum(durs) / max(1, len(durs)) / 1e3, 1)}
    if "TCC_EA0_RDREQ_sum" in c:
        latest["crosscheck_rdreq_x128"] = int(c["TCC_EA0_RDREQ_sum"] * 128)
        latest["crosscheck_wrreq64_x64"] = int(c.get("TCC_EA0_WRREQ_64B_sum", 0) * 64)
    if all(k in c for k in ("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "SQ_WAVES")):
        latest["valu_insts_per_launch"] = int(c["SQ_INSTS_VALU"])
        latest["valu_insts_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
        latest["valu_busy_4clk"] = round(c.get("SQ_ACTIVE_INST_VALU", c["SQ_INSTS_VALU"]) * 4
                                         / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 3)
    json.dump(latest, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1)
    print("PMC " + json.dumps(latest))


if __name__ == "__main__":
    main()
