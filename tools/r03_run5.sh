#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
cp embodied_clip_amd/lib/libec_amd.so /tmp/keep.so
cp ab_libs/prof2.so embodied_clip_amd/lib/libec_amd.so
for ab in 0 8 128 1 2; do
  for B in 256 334 128; do
  EC_CONV_ABLATE=$ab python tools/bench_conv.py --H 14 --Cin 256 --Cout 256 --ks 3 --B $B 2>&1 | grep -v amdgpu
  done
done > $O/ablate.txt
cp /tmp/keep.so embodied_clip_amd/lib/libec_amd.so
cat $O/ablate.txt
