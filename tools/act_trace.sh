cd /tmp && export TMPDIR=/tmp
for N in 32 128; do
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_act$N
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_act$N -o t -- python $GRAFT_REPO_ROOT/tools/bench_act.py --actors $N > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/prof_act$N/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
# last 14 dispatches = one act step
last=rows[-12:]
t0=int(last[0]["Start_Timestamp"])
print("== N=$N")
for r in last:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    g=int(r["Grid_Size_X"])*int(r["Grid_Size_Y"])//max(1,int(r["Workgroup_Size_X"]))
    print("%7.1f us at %7.1f  wgs %4d  %s" % (d, (int(r["Start_Timestamp"])-t0)/1e3, g, r["Kernel_Name"].replace("(anonymous namespace)::","")[:80]))
print("chain", (int(last[-1]["End_Timestamp"])-t0)/1e3)
PY
done
