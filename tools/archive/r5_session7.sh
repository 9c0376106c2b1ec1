#!/bin/bash
# round 5, session 7: the whole default bench line with the new legs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s7
( time timeout 1500 python bench.py > gpurun_out/r5s7/bench.json 2> gpurun_out/r5s7/bench.err ) 2> gpurun_out/r5s7/time.txt
tail -5 gpurun_out/r5s7/bench.err; cat gpurun_out/r5s7/time.txt
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r5s7/bench.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms_per_step", j["ms_per_step"])
print("scale_proxy", json.dumps({k: v for k, v in j.get("scale_proxy", {}).items() if k in ("pageable", "pinned", "vs_value")})[:600])
print("concurrent", json.dumps(j.get("scale_proxy", {}).get("concurrent"))[:1500])
for k, v in j["configs"].items():
    print(k, json.dumps(v.get("to_rgb_hbm"))[:900])
print("per_rank", json.dumps(j["per_rank"]["ranks"][0]))
print("gpu_entropy", json.dumps(j.get("gpu_entropy"))[:800])
PY
