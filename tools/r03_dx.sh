#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_policy.py tests/test_gpu_engine.py tests/test_gpu_configs.py tests/test_gpu_zeroshot.py -x -q -m gpu 2>&1 | tail -2
for v in 0 1 0 1; do echo "DW_T=$v $(EC_DW_TRANSPOSED=$v python tools/bench_update.py --iters 5 | tail -1)"; done
for v in 0 1; do echo "DW_T=$v 32 actors $(EC_DW_TRANSPOSED=$v python tools/bench_update.py --actors 32 --streams 1 --iters 5 | tail -1)"; done
