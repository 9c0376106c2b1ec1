#!/bin/bash
# round 5, session 27: is the plugin's slow mode the heap being trimmed and regrown under a pixel buffer that lies in it?
# (the device reaches the caller's ordinary memory through the page tables the kernel keeps for it; brk moving the end
# of the heap's mapping every frame would throw part of them away every frame)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s27
export JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
export JGA_HUFF_NO_WIDE=1
for seq in P4k P444 L4k,P4k P1080,P4k,P444; do
  echo "-- default malloc" >> gpurun_out/r5s27/modes.txt
  python tools/archive/r5_plugin_modes.py $seq >> gpurun_out/r5s27/modes.txt 2>&1
  echo "-- MALLOC_TRIM_THRESHOLD_=4 GB (the heap is never trimmed)" >> gpurun_out/r5s27/modes.txt
  MALLOC_TRIM_THRESHOLD_=4000000000 python tools/archive/r5_plugin_modes.py $seq >> gpurun_out/r5s27/modes.txt 2>&1
  echo "-- MALLOC_MMAP_THRESHOLD_=1 MB (a frame's pixels get a mapping of their own)" >> gpurun_out/r5s27/modes.txt
  MALLOC_MMAP_THRESHOLD_=1048576 python tools/archive/r5_plugin_modes.py $seq >> gpurun_out/r5s27/modes.txt 2>&1
done
cat gpurun_out/r5s27/modes.txt
