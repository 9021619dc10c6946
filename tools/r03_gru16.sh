#!/bin/bash
# A/B of the round-3 update-phase changes: 16x16-tile GRU step kernels (EC_GRU_FUSED=2 vs 1), re-ordered weight_ih (EC_WIH_PERM=1 vs 0)
cd $GRAFT_REPO_ROOT; O=gpurun_out/gru16; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_policy.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py tests/test_gpu_engine.py tests/test_gpu_zeroshot.py -x -q -m gpu > $O/tests.log 2>&1; tail -5 $O/tests.log
for v in "1 0" "2 0" "2 1" "1 0" "2 1"; do set -- $v; echo "EC_GRU_FUSED=$1 EC_WIH_PERM=$2"; EC_GRU_FUSED=$1 EC_WIH_PERM=$2 timeout 300 python tools/bench_update.py --iters 3 | tail -1; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o u -- python $GRAFT_REPO_ROOT/tools/bench_update.py --iters 3 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/stats_noconv.py $(find $O/prof -name "*kernel_stats.csv" | head -1) 0.5 | head -30
rm -rf $O/prof
