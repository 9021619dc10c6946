#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts (separate --pmc passes): bash tools/calibrate_fetch.sh <out.json>
OUT=${1:-gpurun_out/fetch_calibration.json}
cd /tmp && export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/gpurun_out/pmc_calib; rm -rf $D; mkdir -p $D
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/f -o p -- python $GRAFT_REPO_ROOT/tools/calibrate_fetch.py > $D/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/w -o p -- python $GRAFT_REPO_ROOT/tools/calibrate_fetch.py > $D/w.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, json
rec = json.loads([l for l in open("$D/f.log") if l.startswith("CALIB ")][-1][6:])
def last(counter, d, match):
    f = glob.glob(f"$D/{d}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and match in r["Kernel_Name"]]
    # sum the counter's per-dimension rows of the LAST dispatch of that kernel
    did = rows[-1]["Dispatch_Id"]
    return sum(float(r["Counter_Value"]) for r in rows if r["Dispatch_Id"] == did) * 1024.0     # reported in KiB
out = []
for r in rec:
    fs, ws = last("FETCH_SIZE", "f", r["match"]), last("WRITE_SIZE", "w", r["match"])
    out.append({"kernel": r["kernel"], "read_bytes": r["read"], "FETCH_SIZE_bytes": fs, "read_over_FETCH_SIZE": round(r["read"] / fs, 3),
                "write_bytes": r["write"], "WRITE_SIZE_bytes": ws, "write_over_WRITE_SIZE": round(r["write"] / ws, 3)})
    print(out[-1])
json.dump({"note": "exact algorithmic bytes of single launches (inputs past the Infinity Cache, evicted before each launch) over "
                   "rocprofv3's FETCH_SIZE / WRITE_SIZE: the factor to apply to the counter", "launches": out}, open("$OUT", "w"), indent=1)
PY
rm -rf $D
