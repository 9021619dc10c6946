#!/bin/bash
# same-box A/B of two library builds on the update phase and the trunk
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for v in old new; do
    cp ab_libs/$v.so embodied_clip_amd/lib/libec_amd.so
    echo "$v $(python tools/bench_update.py --iters 3 | tail -1) | $(python tools/bench_trunk.py --batch 256 --iters 10 | grep forward | cut -c1-40)"
  done
done
cp ab_libs/new.so embodied_clip_amd/lib/libec_amd.so
