#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_vit.py -x -q -m gpu > $O/pytest_enc.txt 2>&1
tail -2 $O/pytest_enc.txt
cp embodied_clip_amd/lib/libec_amd.so /tmp/keep.so
for v in npm0 epi; do
  cp ab_libs/$v.so embodied_clip_amd/lib/libec_amd.so
  python tools/bench_shapes.py --B 256 > $O/shapes_b256_$v.txt 2>&1
  python tools/bench_shapes.py --B 128 > $O/shapes_b128_$v.txt 2>&1
  for b in 256 128 32; do python tools/bench_trunk.py --batch $b --iters 10 2>&1 | grep -v "plan_hash\|amdgpu"; done > $O/trunk_$v.txt
  python bench.py --no-weak --no-h2d --no-cpu-baseline --no-traffic --steps 3 2>/dev/null | tail -1 > $O/bench_$v.json
done
B=334 ABL=0 python tools/stamps8.py > $O/stamps_epi.txt 2>&1
cp /tmp/keep.so embodied_clip_amd/lib/libec_amd.so
paste <(grep -v amdgpu $O/shapes_b256_npm0.txt | cut -c1-75) <(grep -v amdgpu $O/shapes_b256_epi.txt | cut -c60-75)
paste <(grep -v amdgpu $O/shapes_b128_npm0.txt | cut -c1-75) <(grep -v amdgpu $O/shapes_b128_epi.txt | cut -c60-75)
cat $O/trunk_npm0.txt $O/trunk_epi.txt
for v in npm0 epi; do python -c "import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_step_union_ms'])"; done
grep "phases\|mean" $O/stamps_epi.txt
