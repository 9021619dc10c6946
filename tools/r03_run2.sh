#!/bin/bash
# V1 of the 8-wave kernel (branch-free CMP segments): parity tests, per-shape timings, stamps; clustered vs pinned pieces
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_vit.py -x -q -m gpu > $O/pytest_enc.txt 2>&1
tail -3 $O/pytest_enc.txt
cp embodied_clip_amd/lib/libec_amd.so /tmp/keep.so
for v in v1 v1pin; do
  cp ab_libs/$v.so embodied_clip_amd/lib/libec_amd.so
  python tools/bench_shapes.py --B 256 > $O/shapes_b256_$v.txt 2>&1
  python tools/bench_shapes.py --B 128 > $O/shapes_b128_$v.txt 2>&1
  B=334 ABL=0 python tools/stamps8.py > $O/stamps_$v.txt 2>&1
  EC_CONV_ABLATE=0 python tools/bench_conv.py --H 14 --Cin 256 --Cout 256 --ks 3 --B 334 2>&1 | grep -v amdgpu >> $O/b334.txt
  for b in 256 128; do python tools/bench_trunk.py --batch $b --iters 10 2>&1 | grep -v "plan_hash\|amdgpu"; done > $O/trunk_$v.txt
done
cp /tmp/keep.so embodied_clip_amd/lib/libec_amd.so
grep -h -v amdgpu $O/shapes_b256_*.txt $O/stamps_*.txt $O/b334.txt $O/trunk_*.txt
