#!/bin/bash
# The GPU check of a round (rounds 4, 5): the GPU test suite, the default bench line, rocprofv3 kernel stats of the ROOFLINE LEG ALONE
# (so that roofline.frac can be recomputed from the tracked summary: VERDICT r3 item 5) and of a short bench run,
# PMC traffic of the fused kernel, per-kernel trace + PMC of the entropy stage, plugin frame times, harness FPS.
# tools/round_check.sh TAG [HEAD]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r4c}; export JGA_HEAD=${2:-unknown}
OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
  tail -4 $OUT/pytest.log
fi
( time timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; tail -3 $OUT/bench.time
cp bench_details.json $OUT/bench_details.json; wc -c $OUT/bench.json
tail -3 $OUT/bench.err
# the roofline leg alone, as bench.py runs it: 0.5 s of untimed launches of the fused RGB kernel on 48 resident 4K
# frames, then 50 launches between two HIP events (the LAST 50 launches of the trace)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/roof -o roof -f csv -- python tools/kbench.py --roofline-leg 3840 2160 420 48 > $OUT/roof.log 2>&1
echo "roofline leg rc=$?"
PB="python bench.py --steps 12 --warmup 3 --batch 128 --no-cpu --no-e2e --no-pack --no-other --no-gpu-entropy --no-configs --no-measure-traffic --scale-proxy 0"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT -o stats -f csv -- $PB > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "stats rc=$?"
timeout 600 python tools/pmc_traffic.py --out $OUT/pmc > $OUT/pmc.log 2>&1; echo "pmc rc=$?"; tail -2 $OUT/pmc.log
cp profiles/pmc_latest.json $OUT/pmc_latest.json 2>/dev/null
rm -f $OUT/*agent_info.csv $OUT/pmc/*agent_info.csv $OUT/roof/*agent_info.csv
# entropy stage: per-kernel trace and PMC
rm -rf $OUT/hprof; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/hprof -o h -f csv -- python tools/hbench.py > $OUT/hbench.txt 2>&1
python3 - <<PY | tee $OUT/huffman_kernels.txt
import csv,glob
fn=glob.glob("$OUT/hprof/**/h_kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(fn)))
print("# tools/hbench.py (48 x 4K 4:2:0 q90) under rocprofv3 --kernel-trace: us per launch, last launches")
for name in ("hj_init","hj_sync_round","hj_sync_sparse","hj_list_build","hj_sync_list","hj_scan","hj_write(","hj_block_starts","hj_write_blocks","hj_dc_scan","hj_dc_apply","jga_idct","fillBuffer"):
    r=[x for x in rows if name in x["Kernel_Name"]]
    if r: print(name,[round((int(x["End_Timestamp"])-int(x["Start_Timestamp"]))/1e3) for x in r][-14:], set((x["VGPR_Count"],x["LDS_Block_Size"]) for x in r))
PY
grep "Mpix/s\|equal" $OUT/hbench.txt | tail -4 >> $OUT/huffman_kernels.txt
rm -rf $OUT/hprof
bash tools/pmc_entropy.sh $TAG/hpmc > $OUT/huffman_pmc.txt 2>&1; rm -rf $OUT/hpmc; grep -A3 "^hj_write\|^hj_sync_round" $OUT/huffman_pmc.txt | head -12
# one frame through the plugin per registration mode, and the harness
python - <<PY > $OUT/plugin_latency.txt 2>&1
import sys, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from jpeg_gpu_amd import abi, lib, synth
import configs_bench
print("# decode_image(RGB) incl. the pixels in img->pixels, ms per frame; jga_plugin_config.register_buffers: 0 per call (default), 1 for the life of the context, -1 staged copies")
for mode in (0, 1, -1):
    pc = abi.jga_plugin_config(C.sizeof(abi.jga_plugin_config), mode, 0, 0, 0)
    lib.check(lib.L.jga_plugin_configure(C.byref(pc)))
    for name, w, h, s, ri in (("1080p", 1920, 1080, "420", 0), ("4k", 3840, 2160, "420", 0), ("4k444", 3840, 2160, "444", 0), ("8k_dri", 7680, 4320, "420", -1)):
        data = synth.synthetic_jpeg(w, h, s, quality=90, seed=5, restart_interval=ri)
        print("register_buffers=%2d %-7s %.3f ms/frame" % (mode, name, configs_bench._plugin(lib, abi, data, 12)["ms_per_frame"]), flush=True)
open("$OUT/4k.jpg", "wb").write(synth.synthetic_jpeg(3840, 2160, "420", quality=90, seed=1234))
open("$OUT/1080p.jpg", "wb").write(synth.synthetic_jpeg(1920, 1080, "420", quality=90, seed=1234))
PY
cat $OUT/plugin_latency.txt
for f in 4k 1080p; do for o in rgb yuv; do for reg in 1 0 -1; do
  echo -n "$f -o $o JPEG_GPU_HIP_REGISTER=$reg: "; JPEG_GPU_HIP_REGISTER=$reg timeout 60 jpeg_gpu_amd/jpeg_gpu_hip -o $o --seconds 2 --check $OUT/$f.jpg 2>&1 | grep FPS | tail -1
done; done; done | tee $OUT/harness_fps.txt
rm -f $OUT/4k.jpg $OUT/1080p.jpg
python - <<PY
import csv,glob
for fn in glob.glob("$OUT/**/stats_kernel_stats.csv", recursive=True) + glob.glob("$OUT/roof/**/roof_kernel_stats.csv", recursive=True):
    print(fn)
    for r in list(csv.DictReader(open(fn)))[:12]:
        print("%-60s %6s calls avg %10.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
