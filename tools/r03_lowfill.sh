#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env $1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d --no-plugin 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["roofline"]["avg_step_union_ms"], d["roofline"]["avg_launch_ms"])')"; }
run A=1
run EC_CONV8_MIN_TILES=25
run EC_CONV8_MIN_TILES=75
run EC_CONV8_MIN_TILES=100
run A=1
run EC_CONV8_MIN_TILES=25
run EC_CONV8_LOWFILL=130
run EC_CONV8_LOWFILL=200
