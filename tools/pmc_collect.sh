#!/bin/bash
# Counter passes (rocprofv3 --pmc with --kernel-trace only: counters are collected in their own runs) over an encoder
# micro-benchmark.  usage: tools/pmc_collect.sh <out-name> <python script + args>
#   tools/pmc_collect.sh trunk_r02 tools/bench_trunk.py --batch 256 --iters 1
#   tools/pmc_collect.sh vit_r02   tools/bench_vit.py   --batch 256 --iters 1
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$NAME
CMD="python $GRAFT_REPO_ROOT/$*"
# (--pmc serialises every dispatch: the engine's stream-concurrency check cannot pass under it and is waived for these passes only)
run() { EC_STREAMS_UNVERIFIED=1 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$PASS -o p -- $CMD > $OUT.$PASS.log 2>&1; }
mkdir -p $OUT
PASS=p1 run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
PASS=p2 run SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD
PASS=p3 run FETCH_SIZE GRBM_GUI_ACTIVE
PASS=p4 run WRITE_SIZE
# a plain kernel trace of the same command (durations, un-perturbed by counters)
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o p -- $CMD > $OUT.kt.log 2>&1
grep -h "plan_hash\|ms/forward" $OUT.kt.log | tail -3
ls $OUT
