"""Prints the non-encoder rows of a rocprofv3 --stats kernel_stats.csv (update phase / act step kernels)."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("conv", "stem", "avgpool")):
        continue
    t = int(r["TotalDurationNs"]) / 1e6
    if t > float(sys.argv[2]) if len(sys.argv) > 2 else 1.0:
        c = int(r["Calls"])
        print(f"{t:8.2f} ms {c:6d} calls {t * 1e3 / c:9.1f} us  min {int(r['MinNs']) / 1e3:8.1f}  " + n.replace("(anonymous namespace)::", "")[:90])
