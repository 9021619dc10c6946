"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel totals and, optionally, the last N dispatches."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = cur.execute("select name, grid_x/workgroup_x, lds_size, (end-start), vgpr_count, accum_vgpr_count from kernels order by start").fetchall()
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    agg[r[0]][0] += 1; agg[r[0]][1] += r[3]
tot = sum(v[1] for v in agg.values())
print(f"{'kernel':90s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>8s} {'%':>5s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k.replace('(anonymous namespace)::','')[:90]:90s} {v[0]:6d} {v[1]/1e6:9.3f} {v[1]/v[0]/1e3:8.1f} {100*v[1]/tot:5.1f}")
for r in rows[-n_last:] if n_last else []:
    print(r[0].replace('(anonymous namespace)::','')[:60], r[1], r[2], f"{r[3]/1e3:.1f}us", r[4], r[5])
