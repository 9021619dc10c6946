#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env $1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d --no-plugin 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["roofline"]["avg_step_union_ms"], d["roofline"]["avg_launch_ms"])')"; }
run EC_CONV8_LOWFILL=100
run EC_CONV8_LOWFILL=0
run EC_CONV8_LOWFILL=100
for v in 0 100 200 400; do echo "LOWFILL=$v lone256: $(EC_CONV8_LOWFILL=$v python tools/bench_trunk.py --batch 256 --iters 20 2>/dev/null | grep -v plan_hash | tail -1)"; done
for v in 0 100; do echo "LOWFILL=$v 128 actors: $(EC_CONV8_LOWFILL=$v python bench.py --steps 2 --warmup 1 --actors 128 --no-cpu-baseline --no-h2d --no-plugin 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"])')"; done
for v in 0 100; do echo "LOWFILL=$v zeroshot: $(EC_CONV8_LOWFILL=$v python bench.py --steps 2 --warmup 1 --encoder zeroshot --no-cpu-baseline --no-h2d --no-plugin 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"])')"; done
