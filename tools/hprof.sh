#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/hprof; rocprofv3 --kernel-trace --stats -d gpurun_out/hprof -o h -f csv -- python tools/hbench.py ${@:-3840 2160 420 48 0} > gpurun_out/hprof_out.txt 2>&1
tail -2 gpurun_out/hprof_out.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/hprof/h_kernel_trace.csv")))
for name in ("hj_init","hj_sync_round","hj_sync_sparse","hj_list_build","hj_sync_list","hj_block_starts","hj_write_blocks","hj_dc_scan","hj_write","hj_scan","jga_idct","fillBuffer"):
    r=[x for x in rows if name in x["Kernel_Name"]]
    print(name,[round((int(x["End_Timestamp"])-int(x["Start_Timestamp"]))/1e3) for x in r][:24], set((x["VGPR_Count"],x["SGPR_Count"],x["LDS_Block_Size"],x["Scratch_Size"]) for x in r))
PY
