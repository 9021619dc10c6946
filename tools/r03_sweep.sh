#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "$1 $2: $(env $1 python bench.py --steps 2 --warmup 1 $2 --no-cpu-baseline --no-h2d --no-plugin 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["roofline"]["avg_step_union_ms"])')"; }
for m in 50 75 100 150 50 75; do run EC_CONV8_MIN_TILES=$m "--encoder vit"; done
for m in 50 75 100; do run EC_CONV8_MIN_TILES=$m "--encoder zeroshot"; done
