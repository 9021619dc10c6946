#!/bin/bash
# A/B of the big-tile configurations (EC_CONV_BIG: 0 = 128x128 4-wave, 1 = conv_igemm8 ping-pong, 2 = plain 256x256
# double buffer, 3 = conv_igemm4 one wave per SIMD) on the compute-bound RN50 layer shapes, B = 256.
cd $GRAFT_REPO_ROOT
for big in ${BIGS:-0 3}; do
  export EC_CONV_BIG=$big
  echo "== EC_CONV_BIG=$big"
  python tools/bench_conv.py --H 14 --Cin 256 --Cout 256 --ks 3
  python tools/bench_conv.py --H 14 --Cin 1024 --Cout 256 --ks 1
  python tools/bench_conv.py --H 14 --Cin 512 --Cout 1024 --ks 1
  python tools/bench_conv.py --H 7 --Cin 512 --Cout 512 --ks 3
  python tools/bench_conv.py --H 7 --Cin 2048 --Cout 512 --ks 1
  python tools/bench_conv.py --H 7 --Cin 512 --Cout 2048 --ks 1 --res 1
  python tools/bench_conv.py --H 7 --Cin 1024 --Cout 2048 --ks 1
  python tools/bench_conv.py --H 28 --Cin 128 --Cout 128 --ks 3
  python tools/bench_conv.py --H 56 --Cin 128 --Cout 128 --ks 3 --pool 1
  python tools/bench_conv.py --H 28 --Cin 256 --Cout 256 --ks 3 --pool 1
  python tools/bench_conv.py --H 14 --Cin 512 --Cout 512 --ks 3 --pool 1
  python tools/bench_conv.py --H 14 --Cin 256 --Cout 256 --ks 3 --B 334
done
