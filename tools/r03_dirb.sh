#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/dirb; mkdir -p $O
EC_CONV8_DIRB=1 timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_fullsize.py -x -q -m gpu -k "trunk or rn50 or fullsize or slic" > $O/tests.log 2>&1; tail -4 $O/tests.log
run() { echo "$1 B=$2 $3: $(env $1 python tools/bench_trunk.py --batch $2 $3 --iters 20 2>/dev/null | grep -v plan_hash | tail -1)"; }
for i in 1 2; do
run EC_CONV8_DIRB=0 256 ""
run EC_CONV8_DIRB=1 256 ""
run EC_CONV8_DIRB=0 128 "--min-tiles 50"
run EC_CONV8_DIRB=1 128 "--min-tiles 50"
done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
EC_CONV8_DIRB=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/p$v -o u -- python $GRAFT_REPO_ROOT/tools/bench_trunk.py --batch 256 --iters 5 > /dev/null 2>&1
echo "== DIRB=$v"; grep "igemm8" $(find $GRAFT_REPO_ROOT/$O/p$v -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-120
rm -rf $GRAFT_REPO_ROOT/$O/p$v
done
