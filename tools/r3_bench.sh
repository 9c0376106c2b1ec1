#!/bin/bash
# round 3: bench contract tests + the new parity tests, then the default bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r3_bench}; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q -k "${KEXPR:-bench or graft or evenly or third_id or irregular or pipeline}" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
( time timeout 1500 python bench.py ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; tail -3 $OUT/bench.time
tail -12 $OUT/bench.err
python3 - <<PY
import json
try:
    d=json.load(open("$OUT/bench.json"))
except Exception as e:
    print("no json:", e); raise SystemExit
print({k:d[k] for k in ("value","value_pinned_ingest","ms_per_step") if k in d})
print("verified", d["config"].get("images_verified"), "kept GB", d["config"].get("outputs_kept_GB"))
print("roofline", {k:d["roofline"].get(k) for k in ("achieved","frac","traffic","kernel_ms_per_launch")}, d["roofline"].get("traffic_provenance"))
for k,v in d.get("e2e",{}).items():
    if isinstance(v,dict): print("e2e",k,v.get("value"))
print("gpu_entropy", d.get("gpu_entropy"))
for k,v in d.get("configs",{}).items():
    if isinstance(v,dict): print("cfg",k,json.dumps({a:v[a] for a in ("to_rgb_hbm","to_host_pixels","device","bit_exact_vs_oracle") if a in v})[:900]); print("   cpu", {a:(b.get("value") if isinstance(b,dict) else b) for a,b in v.get("cpu",{}).items()})
PY
