#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "$1 B=$2: $(env $1 python tools/bench_trunk.py --batch $2 --iters 20 2>/dev/null | grep -v plan_hash | tail -1)"; }
for B in 32 64; do
run A=1 $B
run EC_CONV_RING_NS64=6 $B
run EC_CONV_RING_NS64=8 $B
run EC_CONV_RING_NS128=4 $B
run "EC_CONV_RING_NS64=8 EC_CONV_RING_NS128=4" $B
run A=1 $B
done
