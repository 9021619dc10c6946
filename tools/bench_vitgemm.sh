cd $GRAFT_REPO_ROOT
for cfg in "1 0" "4 0" "4 1"; do set -- $cfg; export EC_CONV_BIG=$1 EC_CONV8_BN128=$2; echo "== BIG=$1 BN128=$2"
 for B in 128 64; do
  python tools/bench_conv.py --H 10 --B $B --Cin 768 --Cout 2304 --ks 1
  python tools/bench_conv.py --H 10 --B $B --Cin 768 --Cout 768 --ks 1 --res 1
  python tools/bench_conv.py --H 10 --B $B --Cin 768 --Cout 3072 --ks 1
  python tools/bench_conv.py --H 10 --B $B --Cin 3072 --Cout 768 --ks 1 --res 1
 done
done
