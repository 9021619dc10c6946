#!/bin/bash
# round-3 baseline: per-shape conv timings, trunk at several batches, halo-traffic ablation of the 8-wave kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
python tools/bench_shapes.py --B 256 > $O/shapes_b256.txt 2>&1
python tools/bench_shapes.py --B 128 > $O/shapes_b128.txt 2>&1
python tools/bench_shapes.py --B 32 > $O/shapes_b32.txt 2>&1
for b in 256 128 64 32; do python tools/bench_trunk.py --batch $b --iters 10 2>&1 | grep -v plan_hash; done > $O/trunk.txt
cp embodied_clip_amd/lib/libec_amd.so /tmp/keep.so
cp ab_libs/prof.so embodied_clip_amd/lib/libec_amd.so
for ab in 0 64 1 2; do
  EC_CONV_ABLATE=$ab python tools/bench_conv.py --H 14 --Cin 256 --Cout 256 --ks 3 --B 256 2>&1 | grep -v amdgpu
  EC_CONV_ABLATE=$ab python tools/bench_conv.py --H 14 --Cin 256 --Cout 256 --ks 3 --B 334 2>&1 | grep -v amdgpu
done > $O/ablate.txt
B=334 ABL=0 python tools/stamps8.py > $O/stamps_0.txt 2>&1
B=334 ABL=64 python tools/stamps8.py > $O/stamps_64.txt 2>&1
cp /tmp/keep.so embodied_clip_amd/lib/libec_amd.so
tail -n 30 $O/*.txt
