#!/bin/bash
# round 5, session 33: the headline (bench.py, headline legs only) with the library of commit 1d10f34 (before the
# 12-bit packs, the clears inside the init launch and the single verdict copy) against the current one, alternating
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s33
PB="python bench.py --no-cpu --no-e2e --no-pack --no-other --no-gpu-entropy --no-configs --no-measure-traffic --scale-proxy 0"
for rep in 1 2 3; do for v in old new; do
  if [ $v = old ]; then export JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_old.so; else unset JGA_LIB_PATH; fi
  timeout 300 $PB 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['value'], d.get('value_pinned_ingest'), d['ms_per_step'], d.get('per_rank', {}).get('ranks', [{}])[0].get('h2d_GBps'), d['roofline']['frac'])" >> gpurun_out/r5s33/ab.txt
done; done
cat gpurun_out/r5s33/ab.txt
