# the batch's shared 12-bit tables (85 KB of LDS per list workgroup) under the PIPELINE, where eight lanes' kernels contend for LDS:
# lighter content (device-bound), the headline recipe through steady_sweep, config 4's shard; default against JGA_HUFF_NO_SHARED_WIDE=1
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for pass in 1 2; do
  for e in "X=1" "JGA_HUFF_NO_SHARED_WIDE=1"; do
    echo "== light 2560 | $e"; env JGA_LIB_PATH=$T $e python tools/light_sweep.py 2560 2>&1 | tail -2
    echo "== shard 128 | $e"; env JGA_LIB_PATH=$T $e python tools/shard_sweep.py 128 2>&1 | tail -2
    echo "== 1080p steady | $e"; env JGA_LIB_PATH=$T $e python tools/steady_sweep.py 1920 1080 420 0 2>&1 | tail -1
  done
done
