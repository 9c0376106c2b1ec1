"""gpurun_out/<tag>/ (tools/archive/r2_check.sh, tools/archive/r3_check.sh, tools/round_check.sh) -> profiles/<name>_rocprof_summary.md,
profiles/<name>_bench.json, profiles/pmc_latest.json (+ round 3: <name>_configs_bench.txt,
<name>_huffman_kernels.txt, <name>_huffman_pmc.md, <name>_harness_fps.txt).
Usage: python tools/summarise_check.py r3c r3"""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else tag
src = os.path.join("gpurun_out", tag)
line = json.load(open(os.path.join(src, "bench.json")))          # the one stdout line (round 6: a <= 4 KB summary)
bench = json.load(open(os.path.join(src, "bench_details.json"))) if os.path.exists(os.path.join(src, "bench_details.json")) else line
rf = bench["roofline"]
out = ["# rocprofv3 summary, %s" % name, ""]
roof = glob.glob(os.path.join(src, "roof", "**", "roof_kernel_trace.csv"), recursive=True)
if roof:
    # round 4: the roofline leg ALONE, so that roofline.frac can be recomputed from this file
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(roof[0]))
         if "jga_idct_rgb_kernel" in r["Kernel_Name"]]
    last = d[-50:]
    alg = rf["algorithmic_bytes_per_launch"]
    own = ""
    try:
        own = [l for l in open(os.path.join(src, "roof.log")) if l.startswith("RESULT ")][0][7:].strip()
    except Exception:
        pass
    stat = lambda v: "%d | %.1f | %.1f | %.1f" % (len(v), sum(v) / len(v), min(v), max(v))
    out += ["## the roofline leg alone", "",
            "Command: `rocprofv3 --kernel-trace --stats -- python tools/kbench.py --roofline-leg 3840 2160 420 48` - bench.py's roofline leg as a "
            "program of its own: the fused dequantise + IDCT + upsample + RGB kernel on 48 resident 3840x2160 4:2:0 frames, 0.5 s of untimed "
            "launches (the clocks settle), 5 more, then 50 launches between two HIP events; nothing else on the device.", "",
            "| `jga_idct_rgb_kernel<1,1,true>` | calls | avg us | min us | max us |", "|---|---|---|---|---|",
            "| the 50 timed launches (the last 50 of the trace) | " + stat(last) + " |",
            "| every launch of the process (warm-up included) | " + stat(d) + " |", "",
            "%d algorithmic bytes per launch (48 x (194 400 blocks x 128 B + 3840 x 2160 x 3 B)) / %.1f us = **%.0f GB/s = %.4f of 8 TB/s** by "
            "rocprofv3's clock over the timed launches; the program's own HIP events around the same 50 launches: %s; bench.py's HIP events over "
            "its 50 launches, un-profiled run of the same build: %.1f us -> %.0f GB/s = %.4f (`roofline.frac`)." % (
                alg, sum(last) / len(last), alg / (sum(last) / len(last)) / 1e3, alg / (sum(last) / len(last)) / 1e3 / 8000, own or "n/a",
                rf["kernel_ms_per_launch"] * 1e3, rf["achieved"], rf["frac"]), ""]
    dc = rf.get("device_copy")
    if dc:
        out += ["Copy ceiling of the same bench run (`roofline.device_copy`): hipMemcpyDtoDAsync of %d bytes each way, %d repetitions, warmed: %.4f ms = "
                "%.0f GB/s read + %.0f GB/s written = %.0f GB/s; a copy kernel of this library on the same buffers (csrc/copy_kernel.hip, by grid size): %s GB/s; "
                "torch copy_ of the same volume: %s GB/s.  `roofline.device_copy_GBps` = the best of them = %.0f." % (dc["bytes_each_way"], dc["reps"], dc["ms"], dc["read_GBps"],
                    dc["write_GBps"], dc["read_GBps"] + dc["write_GBps"], dc.get("kernel_copy_GBps_by_grid"), dc.get("torch_copy_GBps"), rf["device_copy_GBps"]), ""]
out += ["## a short bench run", "",
       "Command: `rocprofv3 --kernel-trace --stats -- python bench.py --steps 12 --warmup 3 --batch 128 --no-cpu --no-e2e "
       "--no-pack --no-other --no-gpu-entropy --no-configs --no-measure-traffic --scale-proxy 0` (tools/round_check.sh; rounds 2-3: "
       "tools/archive/r2_check.sh, r3_check.sh without `--batch`): the end-to-end pipeline legs (12 steps of 128 files through 8 lanes in groups "
       "of 32, pageable then pinned files, plus set-up and warm-up batches) followed by the roofline leg "
       "(launches of the fused kernel on 48 resident images).  Under the pipeline several groups' kernels "
       "share the device, so their averages are longer than alone (hj_* alone: <name>_huffman_kernels.txt).", "",
       "## kernel stats", "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
avg = None
for fn in glob.glob(os.path.join(src, "**", "stats_kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        out.append("| %s | %s | %.2f | %.1f | %.1f | %.1f | %s |" % (
            r["Name"][:80], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3,
            float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3, r["Percentage"]))
        if "jga_idct_rgb_kernel<1, 1, true>" in r["Name"]:
            avg = (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, int(r["Calls"]))
out += ["", "The dominant kernel by time in this command is `jga_idct_rgb_kernel<1,1,true>` (the roofline leg's "
        "launches + one per pipeline batch, the latter sharing the GPU with other lanes' kernels).",
        "rocprof: avg %.1f us over %d calls (min %.1f); bench.py's HIP events over its 50 timed launches, un-profiled "
        "run of the same build: %.1f us -> %.0f GB/s of algorithmic bytes = %.3f of 8 TB/s." % (
            avg[0], avg[2], avg[1], rf["kernel_ms_per_launch"]*1e3, rf["achieved"], rf["frac"])]
pm = os.path.join(src, "pmc_latest.json")
if os.path.exists(pm):
    p = json.load(open(pm))
    out += ["", "## PMC of the fused kernel (tools/pmc_traffic.py: `rocprofv3 --kernel-trace --pmc`, one counter group per pass, "
            "child = `tools/kbench.py --child 3840 2160 420 48`)",
            "- FETCH_SIZE = %.0f, WRITE_SIZE = %.0f (units of 1024 B; FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md)" % (p["fetch_size_kb"], p["write_size_kb"]),
            "- HBM traffic per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 = %d B = %.4f x algorithmic (%d B)" % (
                p["hbm_bytes_per_launch"], p["hbm_bytes_per_launch"]/rf["algorithmic_bytes_per_launch"], rf["algorithmic_bytes_per_launch"]),
            "- cross-check: TCC_EA0_RDREQ*128 B = %d, TCC_EA0_WRREQ_64B*64 B = %d" % (p.get("crosscheck_rdreq_x128", 0), p.get("crosscheck_wrreq64_x64", 0)),
            "- VALU: %d wave-instructions per launch, %.1f per wave, busy %.3f of the SIMD cycles at 4 clk per instruction" % (
                p.get("valu_insts_per_launch", 0), p.get("valu_insts_per_wave", 0), p.get("valu_busy_4clk", 0)),
            "- taken at HEAD %s on %s; kernel avg under PMC %.1f us" % (p.get("head"), p.get("date"), p.get("kernel_avg_us_under_pmc", 0))]
    shutil.copy(pm, os.path.join("profiles", "pmc_latest.json"))
out += ["", "## bench line of the same build (un-profiled run)", "```", json.dumps({k: bench[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step")}), "```",
        "e2e: " + json.dumps({k: v.get("value") for k, v in bench.get("e2e", {}).items() if isinstance(v, dict)}),
        "cpu_baseline: " + json.dumps({k: (v.get("value") if isinstance(v, dict) else v) for k, v in bench.get("cpu_baseline", {}).items() if k in ("value", "kind", "cores", "reference_xjpeg_yuv", "libjpeg_turbo_rgb", "oracle_port_rgb")}),
        "other_kernels: " + json.dumps({k: (v["ms"], v["GBps"]) for k, v in bench.get("other_kernels", {}).items()}),
        "gpu_entropy: " + json.dumps({k: bench["gpu_entropy"][k] for k in ("value", "huffman_ms", "idct_rgb_ms", "sync_rounds")} if "gpu_entropy" in bench else {})]
open(os.path.join("profiles", "%s_rocprof_summary.md" % name), "w").write("\n".join(out) + "\n")
# (round 6) <name>_bench.json = the ONE stdout line as printed, <name>_bench_details.json = the full report beside it
if bench is not line:
    open(os.path.join("profiles", "%s_bench.json" % name), "w").write(json.dumps(line, separators=(",", ":")) + "\n")
    json.dump(bench, open(os.path.join("profiles", "%s_bench_details.json" % name), "w"), indent=1)
else:
    json.dump(bench, open(os.path.join("profiles", "%s_bench.json" % name), "w"), indent=1)
print("\n".join(out))

# round 3 extras
if "configs" in bench:
    with open(os.path.join("profiles", "%s_configs_bench.txt" % name), "w") as f:
        f.write("# bench.py `configs` object of the same run: every BASELINE.json config on one MI355X beside its CPU path\n")
        for k, v in bench["configs"].items():
            f.write("%s %s\n" % (k, json.dumps(v)))
for fn, to in (("huffman_kernels.txt", "%s_huffman_kernels.txt"), ("huffman_pmc.txt", "%s_huffman_pmc.md"),
               ("harness_fps.txt", "%s_harness_fps.txt"), ("plugin_latency.txt", "%s_plugin_latency.txt")):
    if os.path.exists(os.path.join(src, fn)):
        shutil.copy(os.path.join(src, fn), os.path.join("profiles", to % name))
