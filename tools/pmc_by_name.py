"""Aggregate the rocprofv3 passes of tools/pmc_collect.sh BY KERNEL NAME over the whole run (used for the PPO update
phase, where the launch sequence is not one fixed plan): per kernel calls, mean duration, MFMA busy % of busy-CU SIMD
cycles, HBM fetch (FETCH_SIZE x 2 on gfx950) and write MB per call.
usage: python tools/pmc_by_name.py gpurun_out/pmc_<name> [min total ms] > profiles/<...>.txt"""
import collections
import csv
import os
import sys

root = sys.argv[1]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0


def find_csv(d, suffix):
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(base, f)
    return None


def short(n):
    return n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]


cnt = collections.defaultdict(lambda: collections.defaultdict(float))
for p in sorted(os.listdir(root)):
    f = find_csv(os.path.join(root, p), 'counter_collection.csv')
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        cnt[short(r['Kernel_Name'])][r['Counter_Name']] += float(r['Counter_Value'])
dur = collections.defaultdict(lambda: [0, 0.0])
kt = find_csv(os.path.join(root, 'kt'), 'kernel_trace.csv')
for r in csv.DictReader(open(kt)):
    d = dur[short(r['Kernel_Name'])]
    d[0] += 1
    d[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('%-58s %6s %9s %9s | %9s | %9s %9s' % ('kernel', 'calls', 'total ms', 'avg us', 'mfma_busy%', 'fetchMB/call', 'writeMB/call'))
for name, (n, us) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    if us / 1e3 < min_ms:
        continue
    c = cnt.get(name, {})
    busy = 100.0 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / max(1.0, 4.0 * c.get('SQ_BUSY_CU_CYCLES', 0.0))
    print('%-58s %6d %9.2f %9.1f | %9.1f | %9.1f %9.1f' % (name[:58], n, us / 1e3, us / n, busy,
                                                        c.get('FETCH_SIZE', 0.0) * 2 * 1024 / 1e6 / n, c.get('WRITE_SIZE', 0.0) * 1024 / 1e6 / n))
