cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export JGA_HUFF_EXP_NOZERO=1; else unset JGA_HUFF_EXP_NOZERO; fi
  echo -n "nozero=$v :: "; timeout 300 python tools/r3_light_e2e.py 2>&1 | tail -1
done
