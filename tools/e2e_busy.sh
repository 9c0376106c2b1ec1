#!/bin/bash
# Device occupancy of the transport-2 pipeline: union of kernel-busy time vs wall, per-kernel
# totals, copy totals (rocprofv3 kernel + memory-copy trace of tools/e2e_trace.py).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/e2eb; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/e2eb -o h -f csv -- python tools/e2e_trace.py "$@" > gpurun_out/e2eb_out.txt 2>&1
grep images gpurun_out/e2eb_out.txt
python3 - <<PY
import csv, collections
k=[(int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"].split("(")[0][:32]) for x in csv.DictReader(open("gpurun_out/e2eb/h_kernel_trace.csv"))]
c=[(int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Direction"], int(x.get("Size", 0) or 0)) for x in csv.DictReader(open("gpurun_out/e2eb/h_memory_copy_trace.csv"))]
k.sort()
end=max(e for _,e,_ in k); t1=end; t0=end-int(100e6)       # the last 100 ms = steady state of the timed run
kk=[x for x in k if x[0]>=t0]
busy=0; cur_s=None; cur_e=None
for s,e,_ in kk:
    if cur_e is None or s>cur_e:
        if cur_e is not None: busy+=cur_e-cur_s
        cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print("window %.1f ms, kernels busy (union) %.1f ms = %.0f %%" % ((t1-t0)/1e6, busy/1e6, 100*busy/(t1-t0)))
tot=collections.defaultdict(lambda:[0,0])
for s,e,n in kk: tot[n][0]+=e-s; tot[n][1]+=1
for n,(d,cnt) in sorted(tot.items(), key=lambda x:-x[1][0]): print("  %-34s %8.2f ms  %6d launches  avg %8.1f us" % (n, d/1e6, cnt, d/cnt/1e3))
ct=collections.defaultdict(lambda:[0,0,0])
for s,e,d,sz in c:
    if s>=t0: ct[d][0]+=e-s; ct[d][1]+=1; ct[d][2]+=sz
for d,(t,cnt,sz) in ct.items(): print("  copy %-28s %8.2f ms  %6d  %.1f MB" % (d, t/1e6, cnt, sz/1e6))
PY
