#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "$1 actors=$2 streams=$3: $(env $1 python bench.py --steps 2 --warmup 1 --actors $2 --encoder-streams $3 --no-cpu-baseline --no-h2d --no-plugin 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["roofline"].get("avg_step_union_ms"), d["roofline"].get("avg_launch_ms"), d["config"]["encoder_streams"])')"; }
run A=1 32 1
run EC_MIN_ACTORS_SLICED=32 32 2
run A=1 48 1
run EC_MIN_ACTORS_SLICED=32 48 2
run A=1 64 2
run A=1 64 1
