#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do echo "LONGSEG=$v: $(EC_CONV8_LONGSEG=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-h2d --no-plugin 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["roofline"]["avg_step_union_ms"])')"; done
for v in 0 1; do echo "LONGSEG=$v vit: $(EC_CONV8_LONGSEG=$v python bench.py --steps 2 --warmup 1 --encoder vit --no-cpu-baseline --no-h2d --no-plugin 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["roofline"]["avg_step_union_ms"])')"; done
for v in 0 1; do echo "LONGSEG=$v 64 actors: $(EC_CONV8_LONGSEG=$v python bench.py --steps 2 --warmup 1 --actors 64 --no-cpu-baseline --no-h2d --no-plugin 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["roofline"]["avg_step_union_ms"])')"; done
