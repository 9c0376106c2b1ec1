#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/shard_trace.py 128 short_job=2 2>&1 | awk '/==== traced/{f=1} f' > gpurun_out/r4s3_trace_short.txt
timeout 300 python tools/shard_trace.py 128 2>&1 | awk '/==== traced/{f=1} f' > gpurun_out/r4s3_trace_default.txt
timeout 300 python tools/shard_trace.py 128 short_job=2 unstuff=2 pinned=1 2>&1 | awk '/==== traced/{f=1} f' > gpurun_out/r4s3_trace_short_pinned.txt
(timeout 600 python -m pytest tests/test_harness.py tests/test_gpu_parity.py -m gpu -q -x -k "registered or input_cache or plugin or alternate" 2>&1 | tail -5) > gpurun_out/r4s3_pytest.txt
python - <<'PY' > gpurun_out/r4s3_plugin.txt 2>&1
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from jpeg_gpu_amd import abi, lib, synth
import configs_bench, ctypes as C
for mode in (0, 1, -1):
    pc = abi.jga_plugin_config(C.sizeof(abi.jga_plugin_config), mode, 0, 0, 0)
    lib.check(lib.L.jga_plugin_configure(C.byref(pc)))
    for name, w, h, s, ri in (("1080p", 1920, 1080, "420", 0), ("4k", 3840, 2160, "420", 0), ("4k444", 3840, 2160, "444", 0), ("8k_dri", 7680, 4320, "420", -1)):
        data = synth.synthetic_jpeg(w, h, s, quality=90, seed=5, restart_interval=ri)
        r = configs_bench._plugin(lib, abi, data, 12)
        print("register_buffers=%2d %-7s %.3f ms/frame" % (mode, name, r["ms_per_frame"]), flush=True)
PY
cat gpurun_out/r4s3_trace_short.txt | tail -30; echo; tail -22 gpurun_out/r4s3_trace_default.txt; echo; tail -12 gpurun_out/r4s3_trace_short_pinned.txt; cat gpurun_out/r4s3_pytest.txt gpurun_out/r4s3_plugin.txt
