# 256-byte subsequences with the dense kernel in groups of 128 + list rounds (JGA_HUFF_DENSE256, tuning build) against the default (128 bytes)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for cfg in "3840 2160 420 48 0" "1920 1080 420 64 0"; do
 for pass in 1 2; do
  for e in "X=1" "JGA_HUFF_SUB=256 JGA_HUFF_DENSE256=1" "JGA_HUFF_SUB=256 JGA_HUFF_DENSE256=1 JGA_HUFF_LITE=129" "JGA_HUFF_SUB=256 JGA_HUFF_DENSE256=1 JGA_HUFF_LITE=177" "JGA_HUFF_SUB=256 JGA_HUFF_DENSE256=1 JGA_HUFF_LITE=129 JGA_HUFF_ITERS=2,3,6"; do
    echo "== $cfg | $e"
    env JGA_LIB_PATH=$T $e python tools/hbench.py $cfg 2>&1 | grep -E "huffman|equal" | tail -3 | cut -c1-140
  done
 done
done
