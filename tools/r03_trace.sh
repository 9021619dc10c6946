#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/gru16; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
EC_WIH_PERM=$v timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr$v -o u -- python $GRAFT_REPO_ROOT/tools/bench_update.py --iters 1 > $GRAFT_REPO_ROOT/$O/tr$v.log 2>&1
f=$(find $GRAFT_REPO_ROOT/$O/tr$v -name "*kernel_trace.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/$O/trace_perm$v.csv
rm -rf $GRAFT_REPO_ROOT/$O/tr$v
done
