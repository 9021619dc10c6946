#!/bin/bash
# Same-box A/B of library builds:  bash tools/ab.sh OUTDIR "libA.so libB.so ..." "<bench args 1>" "<bench args 2>" ...
# Every (library, args) pair runs bench.py once; the value lines land in gpurun_out/OUTDIR/ab.txt.  Libraries are paths
# relative to the repo root (ab_libs/*.so are builds kept aside by hand; embodied_clip_amd/lib/libec_amd.so = HEAD).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/$1; shift
LIBS=$1; shift
mkdir -p $O
: > $O/ab.txt
for ARGS in "$@"; do
  for L in $LIBS; do
    EC_AMD_LIB=$PWD/$L python bench.py $ARGS --no-cpu-baseline --no-h2d --no-plugin --no-sync-actions --no-traffic > $O/line.json 2> $O/err.txt
    python - "$L" "$ARGS" $O/line.json >> $O/ab.txt <<'PY'
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
