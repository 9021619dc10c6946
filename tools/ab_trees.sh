#!/bin/bash
# Same-box A/B of WHOLE TREES (library + host code):  bash tools/ab_trees.sh OUTDIR "treeA treeB ..." "<bench args 1>" ...
# A tree is a directory holding a built copy of the repository ("." = this one; ab_libs/r05_tree = `git archive` of round 5's
# final commit built in place).  Every (tree, args) pair runs that tree's own bench.py once, alternating trees per args line.
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$PWD
O=$ROOT/gpurun_out/$1; shift
TREES=$1; shift
mkdir -p $O
: > $O/ab.txt
for ARGS in "$@"; do
  for T in $TREES; do
    (cd $ROOT/$T && python bench.py $ARGS --no-cpu-baseline --no-h2d --no-plugin --no-sync-actions --no-traffic > $O/line.json 2> $O/err.txt)
    python - "$T" "$ARGS" $O/line.json >> $O/ab.txt <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
    print(f"{sys.argv[1]:22s} | {sys.argv[2]:40s} | {d['value']:10.1f} | ms/step {d['ms_per_step']:8.2f} | update_ms {d.get('update_ms')} | union {d['roofline'].get('avg_step_union_ms')}")
except Exception as e:
    print(f"{sys.argv[1]:22s} | {sys.argv[2]:40s} | FAILED {e}")
PY
  done
done
cat $O/ab.txt
