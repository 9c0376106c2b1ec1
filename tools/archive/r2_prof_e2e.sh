#!/bin/bash
# kernel stats of the transport-2 pipeline for both unstuff modes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for hu in 0; do
  OUT=gpurun_out/prof_e2e_hu$hu; rm -rf $OUT; mkdir -p $OUT
  UNSTUFF=$((2-hu)) SWEEP_CFGS="32,8,24" timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o s -f csv -- python tools/e2e_sweep2.py 1536 > $OUT/run.txt 2>&1
  echo "host_unstuff=$hu"; tail -1 $OUT/run.txt
  python3 - <<PY
import csv,glob
for fn in glob.glob("$OUT/**/*s_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(fn)))[:16]:
        print("  %-58s %6s calls total %9.2f ms avg %9.1f us" % (r["Name"][:58], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
  rm -f $OUT/*agent_info.csv $OUT/*kernel_trace.csv
done
