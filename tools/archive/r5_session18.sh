#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s18
timeout 600 python tools/configs_bench.py > gpurun_out/r5s18/configs.txt 2> gpurun_out/r5s18/configs.err
grep config4 gpurun_out/r5s18/configs.txt | python -c "
import sys, json
for l in sys.stdin:
    k, v = l.split(' ', 1)
    print(json.dumps(json.loads(v)['to_rgb_hbm']['rank3_shard_of_8']))
"
