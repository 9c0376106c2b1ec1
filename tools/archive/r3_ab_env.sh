#!/bin/bash
# interleaved A/B of environment settings on the bench headline: tools/r3_ab_env.sh "ENV_A" "ENV_B" [reps]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${3:-3}); do for e in "$1" "$2"; do
  echo -n "[$e] "; env $e timeout 300 python bench.py --no-cpu --no-e2e --no-pack --no-other --no-gpu-entropy --no-configs --no-measure-traffic ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_pinned_ingest'])"
done; done
