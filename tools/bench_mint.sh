#!/bin/bash
# per-shape A/B of the 8-wave-kernel tile threshold on the launches that fall between 50 and 150 tiles
cd $GRAFT_REPO_ROOT
run() { for mt in 150 50; do echo -n "mint=$mt  "; EC_CONV8_MIN_TILES=$mt python tools/bench_conv.py "$@" 2>/dev/null | grep -v amdgpu; done; }
run --H 14 --Cin 1024 --Cout 512 --ks 1 --B 64
run --H 14 --Cin 512 --Cout 512 --ks 3 --pool 1 --B 64
run --H 7 --Cin 1024 --Cout 2048 --ks 1 --B 64
run --H 7 --Cin 2048 --Cout 512 --ks 1 --B 128
run --H 7 --Cin 512 --Cout 512 --ks 3 --B 128
run --H 14 --Cin 1024 --Cout 256 --ks 1 --B 128
run --H 14 --Cin 256 --Cout 256 --ks 3 --B 128
