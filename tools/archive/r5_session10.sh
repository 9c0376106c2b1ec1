#!/bin/bash
# round 5, session 10: PMC passes for the 4:4:4 rows kernel, the fused 4:2:0 kernel, the PACK expansion; entropy stage PMC
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s10
bash tools/pmc_kernel.sh jga_idct_rgb_rows r5_pmc_rows -- python tools/kbench.py --roofline-leg 3840 2160 444 24 > gpurun_out/r5s10/pmc_rows.txt 2>&1
bash tools/pmc_kernel.sh jga_idct_rgb_kernel r5_pmc_rgb -- python tools/kbench.py --roofline-leg 3840 2160 420 48 > gpurun_out/r5s10/pmc_rgb420.txt 2>&1
bash tools/pmc_kernel.sh jga_unpack r5_pmc_unpack -- python tools/ubench.py 48 > gpurun_out/r5s10/pmc_unpack.txt 2>&1
cat gpurun_out/r5s10/pmc_rows.txt gpurun_out/r5s10/pmc_rgb420.txt gpurun_out/r5s10/pmc_unpack.txt
bash tools/pmc_hbench.sh r5_pmc_huff > gpurun_out/r5s10/pmc_huff.txt 2>&1; tail -60 gpurun_out/r5s10/pmc_huff.txt
