#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in 32 64; do for s in 1 2 4; do echo "actors=$a streams=$s: $(python bench.py --steps 2 --warmup 1 --actors $a --encoder-streams $s --no-cpu-baseline --no-h2d --no-plugin --phase-times 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d.get("phase_ms"))')"; done; done
