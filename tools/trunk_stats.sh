#!/bin/bash
# per-kernel rocprofv3 statistics of one trunk forward at a given batch: bash tools/trunk_stats.sh 32
B=${1:-256}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_t$B
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_t$B -o t -- python $GRAFT_REPO_ROOT/tools/bench_trunk.py --batch $B --iters 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_t$B/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
n=50
last=rows[-n:]
tot=0
for i,r in enumerate(last):
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3; tot+=d
    g=int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"])
    print("%2d %7.1f us  wgs %5d  lds %6s  %s" % (i, d, g, r.get("LDS_Block_Size","?"), r["Kernel_Name"].replace("(anonymous namespace)::","")[:70]))
print("total", round(tot,1), "us; wall of the forward", (int(last[-1]["End_Timestamp"])-int(last[0]["Start_Timestamp"]))/1e3)
PY
