#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do for f in jpeg_gpu_amd/variants/*.so; do
  echo -n "$(basename $f) "; JGA_LIB_PATH=$PWD/$f timeout 300 python bench.py --no-cpu --no-e2e --no-pack --no-other --no-gpu-entropy --no-configs --no-measure-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_pinned_ingest'])"
done; done
