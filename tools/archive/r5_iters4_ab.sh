cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for cfg in "1920 1080 420 1 0" "3840 2160 420 1 0" "3840 2160 444 1 0" "1920 1080 420 4 0" "1920 1080 grey 1 0" "3840 2160 420 2 0" "1920 1080 444 1 0" "1280 720 420 1 0" "1920 1080 420 1 60"; do
 for pass in 1 2; do
  for e in "X=1" "JGA_HUFF_ITERS=3,3,6"; do
    echo "== $cfg | $e"
    env JGA_LIB_PATH=$T $e python tools/hbench.py $cfg 2>&1 | grep -E "huffman" | tail -1
  done
 done
done
