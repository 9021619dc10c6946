#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/b32; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 32 64; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -o u -- python $GRAFT_REPO_ROOT/tools/bench_trunk.py --batch $B --iters 3 > $GRAFT_REPO_ROOT/$O/tr$B.log 2>&1
cp $(find $GRAFT_REPO_ROOT/$O/tr -name "*kernel_trace.csv" | head -1) $GRAFT_REPO_ROOT/$O/trace_b$B.csv; rm -rf $GRAFT_REPO_ROOT/$O/tr
tail -2 $GRAFT_REPO_ROOT/$O/tr$B.log
done
