#!/bin/bash
# PMC passes over the RN50 trunk micro-bench (counters only, kernel-trace for names).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${1:-trunk}
CMD="python $GRAFT_REPO_ROOT/tools/bench_trunk.py --batch 256 --iters 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p1 -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/p2 -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p3 -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/p4 -o p -- $CMD > /dev/null 2>&1
ls -R $OUT | head -30
