#!/bin/bash
# Device-side timeline of the LAST run of config 4's 128-file shard (uploads and kernels), from rocprofv3's traces.
#   tools/shard_timeline.sh [cfg ...]      e.g. tools/shard_timeline.sh short_job=2 pinned=1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/stl; rm -rf $OUT
cat > /tmp/stl_run.py <<PY
import os, sys, time
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from jpeg_gpu_amd import abi, lib, synth
cfg = eval("dict(%s)" % ",".join(sys.argv[1:])) if len(sys.argv) > 1 else {}
pinned = cfg.pop("pinned", 0)
files = [synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=s) for s in range(16)]
pins = [lib.PinnedBytes(f) for f in files]
src = [p.array for p in pins] if pinned else files
jobs = lib.Pipeline.make_jobs([src[i % 16] for i in range(128)], pinned=bool(pinned))
pl = lib.Pipeline(device=0, nthreads=24, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2, batch=32, depth=8, **cfg)
for _ in range(12):
    pl.run_jobs(jobs)
time.sleep(0.05)
t0 = time.perf_counter(); pl.run_jobs(jobs); print("LAST RUN %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
pl.close()
PY
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT -o h -f csv -- python /tmp/stl_run.py "$@" > gpurun_out/stl_out.txt 2>&1
grep "LAST RUN" gpurun_out/stl_out.txt
python3 - <<PY
import csv, glob, collections
kt = glob.glob("$OUT/**/h_kernel_trace.csv", recursive=True)[0]
ct = glob.glob("$OUT/**/h_memory_copy_trace.csv", recursive=True)[0]
K = [(int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"].split("(")[0].replace("void ", "")[:26], x.get("Stream_Id", x.get("Queue_Id", "?"))) for x in csv.DictReader(open(kt))]
C = [(int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x.get("Direction", ""), x.get("Stream_Id", "?")) for x in csv.DictReader(open(ct))]
end = max(e for s, e, n, q in K)
# the last run: everything within 8 ms before the last kernel's end, starting at the first H2D copy after a gap of 20 ms
ev = sorted([(s, e, "copy " + d.replace("MEMORY_COPY_", "")[:12], q) for s, e, d, q in C] + [(s, e, n, q) for s, e, n, q in K])
last = [x for x in ev if x[0] > end - 12_000_000]
gap = 0
for i in range(1, len(last)):
    if last[i][0] - max(y[1] for y in last[:i]) > 3_000_000: gap = i
last = last[gap:]
t0 = last[0][0]
print("events of the last run: %d; span %.3f ms" % (len(last), (max(x[1] for x in last) - t0) / 1e6))
h2d = [x for x in last if "HOST_TO_DEV" in x[2] and x[1] - x[0] > 20000]
print("H2D copies > 20 us: %d, first starts %.3f, last ends %.3f ms; sum of durations %.3f ms" % (len(h2d), (h2d[0][0] - t0) / 1e6, (max(x[1] for x in h2d) - t0) / 1e6, sum(x[1] - x[0] for x in h2d) / 1e6))
# the link: union of the H2D copies' intervals against the run's span, and every copy of 0.2 MB and more
hv = sorted((s, e) for s, e, n, q in last if "HOST_TO_DEV" in n)
lb, cs, ce = 0, hv[0][0], hv[0][1]
for s, e in hv[1:]:
    if s > ce: lb += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
lb += ce - cs
span = max(x[1] for x in last) - t0
print("link busy (union of H2D copies) %.3f ms = %.2f of the span; until the last byte is over (%.3f ms): %.2f" % (
    lb / 1e6, lb / span, (max(e for s, e in hv) - t0) / 1e6, lb / (max(e for s, e in hv) - t0)))
print("H2D copies of 50 us and more (start ms, duration us, stream):")
print("  " + "  ".join("%.2f/%d/%s" % ((s - t0) / 1e6, (e - s) / 1e3, q) for s, e, n, q in last if "HOST_TO_DEV" in n and e - s > 50000))
per = collections.defaultdict(list)
for s, e, n, q in last:
    if not n.startswith("copy"): per[n].append((s, e))
print("kernel              calls  first start  last end   sum ms")
for n, v in sorted(per.items(), key=lambda kv: kv[1][0][0]):
    print("%-26s %4d  %8.3f  %8.3f  %7.3f" % (n, len(v), (v[0][0] - t0) / 1e6, (max(e for s, e in v) - t0) / 1e6, sum(e - s for s, e in v) / 1e6))
# how much of the span some kernel is running
iv = sorted((s, e) for s, e, n, q in last if not n.startswith("copy"))
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("some kernel running %.3f ms of the span" % (busy / 1e6))
PY
