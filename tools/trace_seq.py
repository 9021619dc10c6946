"""Prints, from a rocprofv3 kernel-trace CSV, every launch of kernels matching a substring with its duration and its two predecessors."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2]
short = lambda n: n.replace("void ", "").split("(")[0][:60]
for i, r in enumerate(rows):
    if pat in r["Kernel_Name"]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        prev = " <- ".join(f"{short(rows[j]['Kernel_Name'])} {(int(rows[j]['End_Timestamp']) - int(rows[j]['Start_Timestamp'])) / 1e3:.0f}us" for j in (i - 1, i - 2) if j >= 0)
        print(f"{d:9.1f} us  grid {r.get('Grid_Size_X', '?'):>8}  {short(r['Kernel_Name'])}   after {prev}")
