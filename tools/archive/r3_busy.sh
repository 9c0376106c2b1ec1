#!/bin/bash
# How busy are the device and the link during the headline stream?  Union of kernel intervals and of
# H2D copy intervals over the last run of tools/e2e_sweep2.py (rocprofv3 kernel + memory-copy traces).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/busy; SWEEP_CFGS="48,8,24" timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/busy -o h -f csv -- python tools/e2e_sweep2.py ${1:-1536} > gpurun_out/busy_out.txt 2>&1
tail -1 gpurun_out/busy_out.txt
python3 - <<PY
import csv, glob
kt = glob.glob("gpurun_out/busy/**/h_kernel_trace.csv", recursive=True)[0]
ct = glob.glob("gpurun_out/busy/**/h_memory_copy_trace.csv", recursive=True)[0]
K = sorted((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"].split("(")[0][:30]) for x in csv.DictReader(open(kt)))
C = sorted((int(x["Start_Timestamp"]), int(x["End_Timestamp"])) for x in csv.DictReader(open(ct)) if "HOST_TO_DEVICE" in x["Direction"])
big = [c for c in C if c[1] - c[0] > 500000]          # the groups' blobs
# the last run = the last 32 big uploads (1536 / 48)
n = 32
last = big[-n:]
t0, t1 = last[0][0], max(k[1] for k in K)
def union(iv):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(iv):
        if e < t0 or s > t1: continue
        s, e = max(s, t0), min(e, t1)
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot
span = t1 - t0
ku = union([(s, e) for s, e, n_ in K])
cu = union(last)
print("span %.1f ms | some kernel running %.1f ms (%.0f %%) | some group upload running %.1f ms (%.0f %%)" % (span/1e6, ku/1e6, 100*ku/span, cu/1e6, 100*cu/span))
import collections
per = collections.defaultdict(lambda: [0, 0])
for s, e, n_ in K:
    if s >= t0: per[n_][0] += e - s; per[n_][1] += 1
for n_, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:8]:
    print("  %-32s %4d launches, %.1f ms in total, avg %.0f us" % (n_, c, t/1e6, t/c/1e3))
print("  uploads: avg %.2f ms each (%.0f MB/s if 148 MB)" % (sum(e - s for s, e in last)/len(last)/1e6, 148e6/(sum(e - s for s, e in last)/len(last)/1e9)/1e6*1e0))
PY
