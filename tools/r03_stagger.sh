#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "1 0" "1 1" "0 0" "0 1" "1 0" "1 1"; do set -- $v; echo "EC_ENGINE_STAGGER=$1 EC_WIH_PERM=$2"; EC_ENGINE_STAGGER=$1 EC_WIH_PERM=$2 timeout 300 python tools/bench_update.py --iters 8 | tail -1; done
