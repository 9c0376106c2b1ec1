#!/bin/bash
# the north-star transport (host Huffman -> dense planes in pinned memory -> DMA -> fused kernel) with the planes leaving
# the host stage through non-temporal stores (default) against plain stores (tuning build, JGA_ENTROPY_NT=0): bench e2e
# leg alone, interleaved on one box
cd /root/repo; mkdir -p gpurun_out
for i in 1 2 3; do for nt in 1 0; do
echo -n "JGA_ENTROPY_NT=$nt "
JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so JGA_ENTROPY_NT=$nt timeout 600 python bench.py --steps 4 --warmup 1 --batch 64 --no-cpu --no-pack --no-other --no-gpu-entropy --no-configs --no-measure-traffic --scale-proxy 0 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:v.get('value') for k,v in b['e2e'].items() if isinstance(v,dict) and ('north' in k or 'pack' in k)})"
done; done > gpurun_out/r4s23_e2e.txt 2>&1
cat gpurun_out/r4s23_e2e.txt
