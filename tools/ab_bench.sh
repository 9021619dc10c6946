#!/bin/bash
# same-box A/B of two builds of the library (alternating runs); BENCH_ARGS="--actors 64" for other operating points.
# Prepare: mkdir ab_libs; build the reference state -> cp embodied_clip_amd/lib/libec_amd.so ab_libs/old.so; build the
# candidate -> cp ... ab_libs/new.so (ab_libs/ is git-ignored and travels with the gpurun snapshot); restore the library after.
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in old new; do
    cp ab_libs/$v.so embodied_clip_amd/lib/libec_amd.so
    python bench.py --no-weak --no-h2d --no-cpu-baseline --no-traffic ${BENCH_ARGS} 2>/dev/null | tail -1 > /tmp/ab.json
    python -c "import json; d=json.load(open('/tmp/ab.json')); print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_step_union_ms'])"
  done
done
