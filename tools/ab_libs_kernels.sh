cd $GRAFT_REPO_ROOT
cp embodied_clip_amd/lib/libec_amd.so /tmp/keep.so
for v in old new old new; do
  cp ab_libs/$v.so embodied_clip_amd/lib/libec_amd.so
  python tools/bench_shapes.py --B 256 --set c3 2>&1 | grep "L3\|L4.0" | sed "s/^/$v /" | cut -c1-100
done
for v in old new old new; do
  cp ab_libs/$v.so embodied_clip_amd/lib/libec_amd.so
  python bench.py --no-weak --no-h2d --no-cpu-baseline --no-traffic --no-plugin --steps 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['avg_step_union_ms'])"
done
cp /tmp/keep.so embodied_clip_amd/lib/libec_amd.so
