# 1080p in a steady state through the pipeline (a long run), interleaved on one box: this build / no shared 12-bit tables / the sparse later rounds of round 4
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for pass in 1 2 3; do
  for e in "X=1" "JGA_HUFF_NO_SHARED_WIDE=1" "JGA_HUFF_LIST=0"; do
    echo -n "$e: "; env JGA_LIB_PATH=$T $e python tools/steady_sweep.py ${GEOM:-1920 1080 420 0} 2>&1 | tail -1 | cut -c42-120
  done
done
