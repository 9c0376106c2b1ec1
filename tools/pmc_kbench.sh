#!/bin/bash
# PMC passes on the library kernels via tools/kbench.py --child (env selects the kernel).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $OUT
CMD="python tools/kbench.py --child 3840 2160 420 32"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT -o a -f csv -- $CMD > /dev/null 2>$OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM -d $OUT -o b -f csv -- $CMD > /dev/null 2>$OUT/b.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o c -f csv -- $CMD > /dev/null 2>$OUT/c.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o d -f csv -- $CMD > /dev/null 2>$OUT/d.err
rocprofv3 --kernel-trace --pmc TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum -d $OUT -o e -f csv -- $CMD > /dev/null 2>$OUT/e.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $OUT -o f -f csv -- $CMD > /dev/null 2>$OUT/f.err
ls $OUT | tr '\n' ' '
