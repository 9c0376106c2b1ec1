#!/bin/bash
# pass 3 alone (jga_yuv_rgb_kernel) with its workgroups per CU capped through dynamic LDS (tuning build): does the copy
# kernel's "16 waves per CU" carry over?
cd /root/repo
for lds in 0 12000 20000 34000 48000 74000; do
echo -n "JGA_YUVRGB_LDS=$lds "
JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so JGA_YUVRGB_LDS=$lds python bench.py --steps 1 --warmup 0 --batch 32 --no-cpu --no-e2e --no-pack --no-gpu-entropy --no-configs --no-measure-traffic --scale-proxy 0 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:(v['ms'],v['GBps']) for k,v in b['other_kernels'].items()})"
done
