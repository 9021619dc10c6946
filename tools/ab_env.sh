#!/bin/bash
# same-box A/B of environment switches: bash tools/ab_env.sh "EC_CONV8_MIN_TILES=150" "EC_CONV8_MIN_TILES=90" ...
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for v in "$@"; do
    env $v python bench.py --no-weak --no-h2d --no-cpu-baseline --no-traffic ${BENCH_ARGS} 2>/dev/null | tail -1 > /tmp/ab.json
    python -c "import json; d=json.load(open('/tmp/ab.json')); print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_step_union_ms'])"
  done
done
