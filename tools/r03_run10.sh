#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest_enc.txt 2>&1
tail -2 $O/pytest_enc.txt
for r in 0 1 2; do
  EC_CONV_ROWSN=$r bash tools/trunk_stats.sh 128 > $O/stats_b128_rowsn$r.txt 2>&1
  EC_CONV_ROWSN=$r bash tools/trunk_stats.sh 256 > $O/stats_b256_rowsn$r.txt 2>&1
  echo "== rowsn=$r"; grep "rows\|total" $O/stats_b128_rowsn$r.txt | cut -c1-90; grep "rows\|total" $O/stats_b256_rowsn$r.txt | cut -c1-90
done
