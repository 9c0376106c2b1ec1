#!/bin/bash
# round 5, session 36: which HIP call holds the one 8-10 ms frame of every process (plugin loop, 4K)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s36
rm -rf gpurun_out/hiptl
timeout 300 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace -d gpurun_out/hiptl -o h -f csv -- python tools/archive/r5_plugin_frames.py 4k none > gpurun_out/r5s36/frames.txt 2>&1
tail -1 gpurun_out/r5s36/frames.txt
ls gpurun_out/hiptl
python3 - > gpurun_out/r5s36/long_calls.txt <<'PY'
import csv, glob
fn = glob.glob("gpurun_out/hiptl/**/h_hip_api_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
print(len(rows), "HIP calls; columns", list(rows[0].keys()))
t0 = int(rows[0]["Start_Timestamp"])
long = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"]) - t0, r["Function"], r.get("Thread_Id")) for r in rows]
# the steady part: after the first 60 % of the calls
cut = sorted(x[1] for x in long)[int(len(long) * 0.5)]
for d, s, f, t in sorted([x for x in long if x[1] > cut], reverse=True)[:15]:
    print("%9.3f ms at +%10.3f ms  %s  thread %s" % (d / 1e6, s / 1e6, f, t))
PY
cat gpurun_out/r5s36/long_calls.txt
