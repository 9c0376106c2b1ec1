#!/bin/bash
# round 5, session 25: which process history puts the plugin's decode_image into its slow mode
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s25
export JGA_LIB_PATH=jpeg_gpu_amd/libjpeg_gpu_amd_tuning.so
timeout 1500 python tools/archive/r5_plugin_modes.py > gpurun_out/r5s25/modes.txt 2>&1
cat gpurun_out/r5s25/modes.txt
