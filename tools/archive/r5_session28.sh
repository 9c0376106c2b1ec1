#!/bin/bash
# round 5, session 28: the plugin's slow mode frame by frame, and the device's timeline in it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s28
export JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
for v in nowide wide; do
  if [ $v = nowide ]; then export JGA_HUFF_NO_WIDE=1; else unset JGA_HUFF_NO_WIDE; fi
  for f in 4k 444 1080; do for gap in none write sleep0; do
    echo -n "$v " >> gpurun_out/r5s28/frames.txt
    python tools/archive/r5_plugin_frames.py $f $gap >> gpurun_out/r5s28/frames.txt 2>&1
  done; done
done
cat gpurun_out/r5s28/frames.txt
export JGA_HUFF_NO_WIDE=1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for f in 4k 444; do
rm -rf gpurun_out/ptl; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/ptl -o h -f csv -- python tools/archive/r5_plugin_frames.py $f none > gpurun_out/r5s28/traced_$f.txt 2>&1
tail -1 gpurun_out/r5s28/traced_$f.txt
python3 - > gpurun_out/r5s28/timeline_$f.txt <<PY
import csv
ev=[]
for x in csv.DictReader(open("gpurun_out/ptl/h_kernel_trace.csv")):
    ev.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"].split("(")[0][:40]))
for x in csv.DictReader(open("gpurun_out/ptl/h_memory_copy_trace.csv")):
    ev.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), "copy " + x.get("Direction", "") ))
ev.sort()
ev = ev[-160:]
t0 = ev[0][0]; prev = t0
for s, e, n in ev:
    print("%9.1f us  +%7.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n))
    prev = e
PY
done
head -90 gpurun_out/r5s28/timeline_4k.txt
