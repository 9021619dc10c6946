#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "$1 B=$2: $(env $1 python tools/bench_trunk.py --batch $2 --iters 20 2>/dev/null | grep -v plan_hash | tail -1)"; }
for B in 32 64 128; do
run A=1 $B
run EC_CONV8_MIN_TILES=100 $B
run EC_CONV8_MIN_TILES=50 $B
run EC_CONV8_MIN_TILES=25 $B
run EC_CONV8_MIN_TILES=12 $B
done
