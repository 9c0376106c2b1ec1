#!/bin/bash
# Round-2 GPU check: the GPU test suite, the default bench line, rocprofv3 kernel stats of a
# short bench run, and the PMC traffic passes.  tools/r2_check.sh TAG [HEAD]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r2a}; export JGA_HEAD=${2:-unknown}
OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
  tail -5 $OUT/pytest.log
fi
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -3 $OUT/bench.err
PB="python bench.py --steps 12 --warmup 3 --no-cpu --no-e2e --no-pack --no-other --no-gpu-entropy"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT -o stats -f csv -- $PB > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "stats rc=$?"
timeout 600 python tools/pmc_traffic.py --out $OUT/pmc > $OUT/pmc.log 2>&1; echo "pmc rc=$?"; tail -2 $OUT/pmc.log
cp profiles/pmc_latest.json $OUT/pmc_latest.json 2>/dev/null
rm -f $OUT/*agent_info.csv $OUT/pmc/*agent_info.csv
python - <<PY
import csv,glob
for fn in glob.glob("$OUT/**/stats_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(fn)))[:14]:
        print("%-60s %6s calls avg %10.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
cat $OUT/bench.json
