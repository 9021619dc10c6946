#!/bin/bash
# NPM0 variants (group 0's pieces in MEM0) + end-to-end bench + policy/engine tests after the ABI changes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
cp embodied_clip_amd/lib/libec_amd.so /tmp/keep.so
for v in npm0 npm4 npm8; do
  cp ab_libs/$v.so embodied_clip_amd/lib/libec_amd.so
  python tools/bench_shapes.py --B 256 > $O/shapes_b256_$v.txt 2>&1
  python tools/bench_shapes.py --B 128 > $O/shapes_b128_$v.txt 2>&1
  B=334 ABL=0 python tools/stamps8.py > $O/stamps_$v.txt 2>&1
  python bench.py --no-weak --no-h2d --no-cpu-baseline --no-traffic --steps 3 2>/dev/null | tail -1 > $O/bench_$v.json
done
cp ab_libs/npm0.so embodied_clip_amd/lib/libec_amd.so
timeout 1200 python -m pytest tests/test_gpu_policy.py tests/test_gpu_edges.py tests/test_gpu_engine.py tests/test_gpu_zeroshot.py tests/test_gpu_golden.py -x -q -m gpu > $O/pytest_pol.txt 2>&1
cp /tmp/keep.so embodied_clip_amd/lib/libec_amd.so
tail -3 $O/pytest_pol.txt
for v in npm0 npm4 npm8; do python -c "import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_step_union_ms'])"; done
grep -h "mean cycles" $O/stamps_*.txt
