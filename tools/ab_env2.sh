#!/bin/bash
# same-box A/B of environment switches, alternating: VAR=name VALS="a b c" [BENCH_ARGS=...] bash tools/ab_env2.sh
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for v in $VALS; do
    env $VAR=$v python bench.py --no-weak --no-h2d --no-cpu-baseline --no-traffic --no-plugin --steps 3 ${BENCH_ARGS} 2>/dev/null | tail -1 > /tmp/ab.json
    python -c "import json; d=json.load(open('/tmp/ab.json')); print('$VAR=$v', d['value'], d['ms_per_step'], d['roofline']['avg_step_union_ms'])"
  done
done
