#!/bin/bash
# round 6, session 14: the shard with spinning waits (jga_pipeline_config.spin_waits = 1) against the poll + sleep default
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_s14; mkdir -p $O
for rep in 1 2 3; do
  timeout 300 python tools/shard_sweep.py 128 "" "spin_waits=1" >> $O/shard.txt 2>&1
done
cat $O/shard.txt
