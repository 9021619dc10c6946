"""Per-launch table of the LAST forward in a rocprofv3 kernel-trace CSV of tools/bench_trunk.py (launches per forward given): duration, gap to predecessor."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
rows = rows[-n:]
tot = gaps = 0.0
for i, r in enumerate(rows):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g = (int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3 if i else 0.0
    tot += d; gaps += g
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:70]
    print(f"{i:3d} {d:7.1f} us  gap {g:5.1f}  grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):5d} wg  lds {int(r['LDS_Block_Size']) // 1024:3d}K  {name}")
print(f"kernel time {tot:.1f} us, gaps {gaps:.1f} us, span {(int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3:.1f} us")
