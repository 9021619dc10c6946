#!/bin/bash
# same-box A/B of two builds of the library: ab_libs/old.so vs ab_libs/new.so (alternating runs); BENCH_ARGS="--actors 64"
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in old new; do
    cp ab_libs/$v.so embodied_clip_amd/lib/libec_amd.so
    python bench.py --no-weak --no-h2d --no-cpu-baseline --no-traffic ${BENCH_ARGS} 2>/dev/null | tail -1 > /tmp/ab.json
    python -c "import json; d=json.load(open('/tmp/ab.json')); print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_step_union_ms'])"
  done
done
