#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "$1 streams=$2: $(env $1 python bench.py --steps 2 --warmup 1 --encoder-streams $2 --no-cpu-baseline --no-h2d --no-plugin 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["roofline"].get("avg_step_union_ms"), d["roofline"].get("avg_launch_ms"))')"; }
run A=1 2
run A=1 3
run EC_CONV8_MIN_TILES=30 3
run EC_CONV8_MIN_TILES=50 3
run A=1 4
run EC_CONV8_MIN_TILES=24 4
run A=1 2
