#!/bin/bash
cd $GRAFT_REPO_ROOT
export EC_CONV_BIG=${BIG:-3}
for ab in 0 1 2 4 3 5 6 7 16; do
  EC_CONV_ABLATE=$ab python tools/bench_conv.py --H 14 --Cin 256 --Cout 256 --ks 3 --B 334 2>&1 | grep -v amdgpu
done
