#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -m gpu -q -x -k "pipeline or config4 or short_job or input_cache" 2>&1 | tail -3) > gpurun_out/r4s20.txt
timeout 600 python tools/shard_sweep.py 128 "" "link_slots=-3" >> gpurun_out/r4s20.txt 2>&1
timeout 600 python tools/shard_sweep.py 1024 "" "link_slots=-3" >> gpurun_out/r4s20.txt 2>&1
for ls in 0 -3 0 -3; do echo -n "4K headline, link_slots=$ls: "; timeout 300 python - <<PY
import sys, time
sys.path.insert(0, ".")
import torch
from jpeg_gpu_amd import abi, lib, synth
files = [synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1234 + i) for i in range(16)]
n = 2048
out = lib.DeviceBuffer(((3840*2160*3 + 255)//256*256) * 256)
pitch = (3840*2160*3 + 255)//256*256
pl = lib.Pipeline(device=0, nthreads=24, out=abi.JPEG_DECODE_RGB, copy_back=False, transport=2, batch=32, depth=8, link_slots=$ls)
jobs = lib.Pipeline.make_jobs([files[i % 16] for i in range(n)], dev_outs=[out.ptr + (i % 256) * pitch for i in range(n)])
pl.run_jobs(jobs); pl.run_jobs(jobs)
ts = []
for _ in range(4):
    t0 = time.perf_counter(); rc = pl.run_jobs(jobs); ts.append(time.perf_counter() - t0); assert rc == 0
print(" ".join("%.1f" % (n * 3840 * 2160 / t / 1e9) for t in ts), "Gpixel/s")
pl.close()
PY
done >> gpurun_out/r4s20.txt 2>&1
cat gpurun_out/r4s20.txt
