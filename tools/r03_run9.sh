#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d --no-plugin --no-traffic > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/prof_bench
tail -1 $O/bench_under_rocprof.log | cut -c1-200
