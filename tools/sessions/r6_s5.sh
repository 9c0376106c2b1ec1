#!/bin/bash
# round 6, session 5: GPU suite after the policy changes + hardening tests; the policy sweep again; list statistics and
# first-launch steps on photograph-like content
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_s5; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
export JGA_LIB_PATH=$PWD/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
timeout 1500 python tools/policy_sweep.py > $O/policy_alone.txt 2> $O/policy_alone.err; tail -45 $O/policy_alone.txt | cut -c1-150
timeout 1500 python tools/policy_sweep.py --shared > $O/policy_shared.txt 2> $O/policy_shared.err; tail -3 $O/policy_shared.txt | cut -c1-150
for it in "" "4,3,6" "6,3,6" "8,3,6"; do
  for content in photo recipe; do
    echo "== $content ITERS=$it" >> $O/photo_iters.txt
    for rep in 1 2; do env CONTENT=$content ${it:+JGA_HUFF_ITERS=$it} timeout 200 python tools/hbench.py 3840 2160 420 48 0 2>&1 | grep huffman | tail -1 >> $O/photo_iters.txt; done
  done
done
env CONTENT=photo JGA_HUFF_LIST_STATS=1 timeout 200 python tools/hbench.py 3840 2160 420 48 0 2>&1 | grep "list round" | tail -6 >> $O/photo_iters.txt
env JGA_HUFF_LIST_STATS=1 timeout 200 python tools/hbench.py 3840 2160 420 48 0 2>&1 | grep "list round" | tail -6 >> $O/photo_iters.txt
cat $O/photo_iters.txt
