#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
cp embodied_clip_amd/lib/libec_amd.so /tmp/keep.so
cp ab_libs/npm0.so embodied_clip_amd/lib/libec_amd.so
for B in 256 128; do
EC_CONV_BIG=4 python tools/bench_shapes.py --B $B > $O/shapes_b${B}_big4.txt 2>&1
EC_CONV_BIG=4 EC_CONV8_BN128=1 python tools/bench_shapes.py --B $B > $O/shapes_b${B}_big4_bn128.txt 2>&1
done
for b in 256 128 64 32; do python tools/bench_trunk.py --batch $b --iters 10 2>&1 | grep -v "plan_hash\|amdgpu"; done > $O/trunk.txt
python bench.py --encoder vit --no-weak --no-h2d --no-cpu-baseline --no-traffic --steps 2 2>/dev/null | tail -1 > $O/bench_vit.json
cp /tmp/keep.so embodied_clip_amd/lib/libec_amd.so
grep -h -v amdgpu $O/shapes_b256_big4.txt $O/shapes_b256_big4_bn128.txt $O/trunk.txt
python -c "import json; d=json.load(open('$O/bench_vit.json')); print('vit', d['value'], d['ms_per_step'], d['roofline']['avg_step_union_ms'])"
