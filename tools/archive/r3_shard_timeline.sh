#!/bin/bash
# Device-side timeline of the LAST run of a short job (uploads and kernels per queue), from rocprofv3's traces
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/stl; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/stl -o h -f csv -- python tools/r3_shard_runs.py ${1:-128} > gpurun_out/stl_out.txt 2>&1
tail -3 gpurun_out/stl_out.txt
python3 - <<PY
import csv, glob, collections
kt = glob.glob("gpurun_out/stl/**/h_kernel_trace.csv", recursive=True)[0]
ct = glob.glob("gpurun_out/stl/**/h_memory_copy_trace.csv", recursive=True)[0]
K = [(int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"].split("(")[0][:28], x.get("Stream_Id", x.get("Queue_Id", "?"))) for x in csv.DictReader(open(kt))]
C = [(int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x.get("Direction", ""), x.get("Stream_Id", "?")) for x in csv.DictReader(open(ct))]
big = sorted(c for c in C if "HOST_TO_DEVICE" in c[2] and c[1] - c[0] > 100000)     # (the trace has no byte counts: the groups' blobs are the long ones)
last = big[-8:]
t0 = last[0][0]
print("uploads of the last run (ms from the first one's start):")
for s, e, d, b in last:
    print("  stream %s: %.3f .. %.3f" % (b, (s - t0) / 1e6, (e - t0) / 1e6))
per = collections.defaultdict(list)
for s, e, n, q in K:
    if s >= t0 - 200000:
        per[q].append((s, e, n))
print("kernels per queue: first start, last end, busy ms, dense rounds, sparse, write, idct")
for q, v in sorted(per.items(), key=lambda kv: kv[1][0][0]):
    v.sort()
    busy = sum(e - s for s, e, n in v) / 1e6
    def first(name):
        r = [s for s, e, n in v if name in n]
        return "%.3f" % ((r[0] - t0) / 1e6) if r else "-"
    def lastend(name):
        r = [e for s, e, n in v if name in n]
        return "%.3f" % ((r[-1] - t0) / 1e6) if r else "-"
    print("  q%s: %.3f .. %.3f busy %.3f | first round %s write %s..%s idct %s..%s (%d kernels)" % (
        q, (v[0][0] - t0) / 1e6, (v[-1][1] - t0) / 1e6, busy, first("hj_sync_round"), first("hj_write"), lastend("hj_write"),
        first("jga_idct"), lastend("jga_idct"), len(v)))
PY
