#!/bin/bash
# round 5, session 26: the runtime's own log of the copies of a plugin frame in the slow and in the fast mode
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s26
export JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
cat > /tmp/few.py <<'PY'
import sys, time, os
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from jpeg_gpu_amd import abi, lib, synth
f = synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1234)
with lib.Decoder(f) as d:
    d.read_header(); d.init_image(); d.decode(abi.JPEG_DECODE_RGB)
    for k in range(8):
        sys.stderr.write("=== frame %d\n" % k); sys.stderr.flush()
        t0 = time.perf_counter(); d.reset(); d.read_header(); d.decode(abi.JPEG_DECODE_RGB); dt = time.perf_counter() - t0
        sys.stderr.write("=== frame %d took %.3f ms\n" % (k, dt * 1e3)); sys.stderr.flush()
PY
for v in wide nowide; do
  if [ $v = nowide ]; then export JGA_HUFF_NO_WIDE=1; else unset JGA_HUFF_NO_WIDE; fi
  python /tmp/few.py 2>&1 | grep "took" > gpurun_out/r5s26/plain_$v.txt
  AMD_LOG_LEVEL=4 timeout 300 python /tmp/few.py > /tmp/log_$v.txt 2>&1
  grep -n "=== frame 5" /tmp/log_$v.txt | head -2
  awk '/=== frame 5$/{p=1} p{print} /=== frame 5 took/{p=0}' /tmp/log_$v.txt | cut -c1-260 | head -400 > gpurun_out/r5s26/frame5_$v.txt
  grep "took" /tmp/log_$v.txt > gpurun_out/r5s26/logged_$v.txt
done
cat gpurun_out/r5s26/plain_*.txt
wc -l gpurun_out/r5s26/frame5_*.txt
