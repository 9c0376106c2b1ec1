#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + HBM PMC passes.
# Outputs under gpurun_out/prof_$1/ ; summarised into profiles/ by tools/summarise_profiles.py
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r1}; OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
BATCH=${2:-48}
timeout 900 python bench.py --batch $BATCH > $OUT/bench.json 2> $OUT/bench.err
PB="python bench.py --batch $BATCH --steps 20 --warmup 3 --no-cpu --no-e2e"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o stats -f csv -- $PB > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch -f csv -- $PB > /dev/null 2> $OUT/fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write -f csv -- $PB > /dev/null 2> $OUT/write.err
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $OUT -o tcc -f csv -- $PB > /dev/null 2> $OUT/tcc.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT -o sq -f csv -- $PB > /dev/null 2> $OUT/sq.err
rm -f $OUT/*agent_info.csv
ls $OUT | tr '\n' ' '; cat $OUT/bench.json
