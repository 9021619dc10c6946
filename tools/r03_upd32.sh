#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/upd32; mkdir -p $O
python tools/bench_update.py --actors 32 --streams 1 --iters 5 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o u -- python $GRAFT_REPO_ROOT/tools/bench_update.py --actors 32 --streams 1 --iters 3 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/stats_noconv.py $(find $O/prof -name "*kernel_stats.csv" | head -1) 0.2 | head -32
grep "igemm8" $(find $O/prof -name "*kernel_stats.csv" | head -1) | cut -c1-150 | head -4
rm -rf $O/prof
