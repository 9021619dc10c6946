cd $GRAFT_REPO_ROOT
for cfg in "0 0" "4 1"; do set -- $cfg; export EC_CONV_BIG=$1 EC_CONV8_BN128=$2; echo "== BIG=$1 BN128=$2"
  python tools/bench_conv.py --H 28 --Cin 128 --Cout 128 --ks 3
  python tools/bench_conv.py --H 56 --Cin 128 --Cout 128 --ks 3 --pool 1
  python tools/bench_conv.py --H 7 --Cin 512 --Cout 512 --ks 3
  python tools/bench_conv.py --H 7 --Cin 2048 --Cout 512 --ks 1
  python tools/bench_conv.py --H 28 --Cin 128 --Cout 128 --ks 3 --B 334
done
