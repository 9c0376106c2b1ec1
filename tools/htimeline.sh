#!/bin/bash
# Kernel + memcpy timeline of ONE decode of the GPU entropy stage (last repetition of
# tools/hbench.py): start offset, duration, gap to the previous activity.
# Usage: tools/htimeline.sh W H sampling nimages restart_interval
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/htl; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/htl -o h -f csv -- python tools/hbench.py "$@" > gpurun_out/htl_out.txt 2>&1
tail -2 gpurun_out/htl_out.txt
python3 - <<PY
import csv
ev=[]
for x in csv.DictReader(open("gpurun_out/htl/h_kernel_trace.csv")):
    ev.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"].split("(")[0][:40]))
try:
    for x in csv.DictReader(open("gpurun_out/htl/h_memory_copy_trace.csv")):
        ev.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), "copy " + x.get("Direction", "")))
except FileNotFoundError:
    pass
ev.sort()
# last decode = from the last hj_sync_round back to the copies just before it, to the last jga_idct
last = max(i for i, e in enumerate(ev) if "hj_sync_round" in e[2])
first = last
while first > 0 and ev[first][0] - ev[first - 1][1] < 200000 and "jga_idct" not in ev[first - 1][2]:
    first -= 1
t0 = ev[first][0]; prev = t0
for s, e, n in ev[first:]:
    print("%9.1f us  +%7.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n))
    prev = e
PY
