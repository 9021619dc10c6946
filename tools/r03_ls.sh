#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/ls; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_policy.py tests/test_gpu_fullsize.py tests/test_gpu_vit.py -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
run() { echo "$1 B=$2 $3: $(env $1 python tools/bench_trunk.py --batch $2 $3 --iters 20 2>/dev/null | grep -v plan_hash | tail -1)"; }
for i in 1 2; do
run EC_CONV8_LONGSEG=0 256 ""
run EC_CONV8_LONGSEG=1 256 ""
run EC_CONV8_LONGSEG=0 128 "--min-tiles 50"
run EC_CONV8_LONGSEG=1 128 "--min-tiles 50"
done
for v in 0 1 0 1; do echo "LONGSEG=$v $(EC_CONV8_LONGSEG=$v python tools/bench_update.py --iters 5 | tail -1)"; done
for v in 0 1; do echo "LONGSEG=$v vit $(EC_CONV8_LONGSEG=$v python tools/bench_vit.py --batch 128 --min-tiles 50 --iters 20 2>/dev/null | grep -v plan_hash | tail -1)"; done
