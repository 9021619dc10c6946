"""One env step of the engine from a rocprofv3 kernel-trace CSV: every kernel between two consecutive stem launches
(act step + encoder forward), with start offsets, durations and idle gaps -- what the step period is made of.

  rocprofv3 --kernel-trace --output-format csv -d out -o t -- python bench.py --actors 32 --steps 1 --warmup 1 ...
  python tools/trace_step.py out/.../t_kernel_trace.csv [step_index_from_end=40]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 40
stems = [i for i, r in enumerate(rows) if "stem_conv1" in r["Kernel_Name"]]
i0, i1 = stems[-back - 1], stems[-back]
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:64]
t0 = int(rows[i0]["Start_Timestamp"])
busy, last_end = 0.0, t0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {gap:6.1f}  grid {r.get('Grid_Size_X', '?'):>8} wg {r.get('Workgroup_Size_X', '?'):>4}  {short(r['Kernel_Name'])}")
    busy += (e - s) / 1e3
    last_end = max(last_end, e)
period = (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3
print(f"step period {period:.1f} us, kernel time {busy:.1f} us, {i1 - i0} launches")
