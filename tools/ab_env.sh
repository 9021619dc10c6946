#!/bin/bash
# Same-box A/B of environment switches:  bash tools/ab_env.sh OUTDIR "ENV_A ENV_B ..." "<bench args 1>" ...
# Each ENV is one VAR=value (use "X=0" for a no-op baseline); every (env, args) pair runs bench.py once.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/$1; shift
ENVS=$1; shift
mkdir -p $O
: > $O/ab.txt
for ARGS in "$@"; do
  for E in $ENVS; do
    env $E python bench.py $ARGS --no-cpu-baseline --no-h2d --no-plugin --no-sync-actions --no-traffic > $O/line.json 2> $O/err.txt
    python - "$E" "$ARGS" $O/line.json >> $O/ab.txt <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[3]))
    print(f"{sys.argv[1]:40s} | {sys.argv[2]:40s} | {d['value']:10.1f} | ms/step {d['ms_per_step']:8.2f} | update_ms {d.get('update_ms')} | union {d['roofline']['avg_step_union_ms']}")
except Exception as e:
    print(f"{sys.argv[1]:40s} | {sys.argv[2]:40s} | FAILED {e}")
PY
  done
done
cat $O/ab.txt
