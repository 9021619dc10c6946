"""Join the rocprofv3 passes of tools/pmc_collect.sh per dispatch of the LAST encoder forward and write
  <out>_per_kernel.txt   per-launch table: duration, MFMA-busy %, VALU/LDS mix, HBM bytes
  <out>_hbm_traffic.json the record bench.py reads for roofline.traffic (keyed by the library's launch-plan hash)

usage: python tools/pmc_summary.py gpurun_out/pmc_<name> <launches per forward> <frames per forward> <out prefix> [plan_hash] [alg MB/frame]

HBM bytes = FETCH_SIZE x 2 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B -- MI355X_MICROARCH.md "HBM") + WRITE_SIZE,
both reported by rocprofv3 in KB.  MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): the share of
SIMD-cycles of busy CUs in which the matrix pipe is executing; next to it the same numerator over ALL 1024 SIMDs for the
kernel's wall time (GRBM_GUI_ACTIVE), which also charges idle CUs (tile quantisation) to the kernel.
"""
import collections
import csv
import json
import os
import sys

# FETCH_SIZE factor per kernel family, MEASURED on known byte counts (tools/calibrate_fetch.sh -> profiles/r03_fetch_calibration.json):
# the counter tallies a 128-B request as 64 B, so the factor is 2 for 16-B-per-lane streaming loads (global_load_dwordx4 and
# buffer_load ... lds alike: 1.94 measured on conv_igemm8) and smaller where part of the traffic is narrower requests
FETCH_FACTOR = {'stem_conv1': 1.25, 'conv3x3_rows': 1.40}
FETCH_FACTOR_DEFAULT = 2.0


def fetch_factor(name):
    for k, v in FETCH_FACTOR.items():
        if k in name:
            return v
    return FETCH_FACTOR_DEFAULT


root, n_last, frames, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
plan_hash = sys.argv[5] if len(sys.argv) > 5 else None
alg_mb = float(sys.argv[6]) if len(sys.argv) > 6 else 45.7
KEYS = ('conv_igemm', 'bneck23', 'conv3x3_img', 'stem', 'avgpool', 'conv1x1_pair', 'conv1x1_regw', 'conv3x3_narrow', 'conv3x3_rows', 'layernorm',
        'mha_', 'patchify', 'assemble', 'attn', 'resize')


def find_csv(d, suffix):
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(base, f)
    return None


FIRST = ('stem_conv1', 'stem7_pool', 'patchify_kernel')   # the first launch of a forward


def last_forward(items, name_of):
    """The launches of the LAST forward: the last n_last matching launches, or (n_last == 0) everything from the last first-of-a-forward
    launch on -- the launch count of a plan depends on the batch (ops that emit a second output skip the launch that would have made it)."""
    if n_last > 0:
        return items[-n_last:]
    starts = [i for i, it in enumerate(items) if any(k in name_of(it) for k in FIRST)]
    return items[starts[-1]:] if starts else items


per = collections.OrderedDict()
for p in sorted(os.listdir(root)):
    f = find_csv(os.path.join(root, p), 'counter_collection.csv')
    if not f:
        continue
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        d = int(r['Dispatch_Id'])
        disp.setdefault(d, {'name': r['Kernel_Name'], 'grid': int(r['Grid_Size']) // max(1, int(r['Workgroup_Size']))})
        disp[d][r['Counter_Name']] = disp[d].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    ids = last_forward([d for d in disp if any(k in disp[d]['name'] for k in KEYS)], lambda d: disp[d]['name'])
    for i, d in enumerate(ids):
        per.setdefault(i, {}).update(disp[d])
# durations from the plain kernel trace
kt = find_csv(os.path.join(root, 'kt'), 'kernel_trace.csv')
if kt:
    rows = last_forward([r for r in csv.DictReader(open(kt)) if any(k in r['Kernel_Name'] for k in KEYS)], lambda r: r['Kernel_Name'])
    for i, r in enumerate(rows):
        per.setdefault(i, {})['dur_us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        per[i].setdefault('name', r['Kernel_Name'])

lines = ['idx kernel                               grid |  dur_us | mfma_busy%%(busy CUs) mfma%%(all SIMDs,wall) | wait_any%% active%% valu%% | valu/mfma lds/mfma bankconf | fetchMiB(calibrated) writeMiB']
tot_f = tot_w = tot_us = 0.0
agg = collections.OrderedDict()
for i, r in per.items():
    wc = r.get('SQ_WAVE_CYCLES', 0) or 1
    nm = r.get('name', '?').replace('void ', '').replace('(anonymous namespace)::', '')
    nm = nm.split('(')[0][:36]
    mb = r.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    busy_cu = r.get('SQ_BUSY_CU_CYCLES', 0.0)
    gui = r.get('GRBM_GUI_ACTIVE', 0.0)
    u1 = 100 * mb / (4 * busy_cu) if busy_cu else float('nan')
    u2 = 100 * mb / (1024 * gui) if gui else float('nan')
    fmb, wmb = fetch_factor(r.get('name', '')) * r.get('FETCH_SIZE', 0) / 1024, r.get('WRITE_SIZE', 0) / 1024
    tot_f += fmb; tot_w += wmb; tot_us += r.get('dur_us', 0)
    a = agg.setdefault(nm, [0, 0.0, 0.0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += r.get('dur_us', 0); a[2] += mb; a[3] += busy_cu; a[4] += fmb; a[5] += wmb
    lines.append(f"{i:3d} {nm:36s} {r.get('grid', 0):5d} | {r.get('dur_us', 0):7.1f} | {u1:8.1f} {u2:8.1f} | "
                 f"{100 * r.get('SQ_WAIT_ANY', 0) / wc:5.1f} {100 * r.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.1f} "
                 f"{100 * r.get('SQ_ACTIVE_INST_VALU', 0) / wc:5.1f} | {r.get('SQ_INSTS_VALU', 0) / max(1, r.get('SQ_INSTS_MFMA', 1)):6.1f} "
                 f"{r.get('SQ_INSTS_LDS', 0) / max(1, r.get('SQ_INSTS_MFMA', 1)):5.1f} "
                 f"{r.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, r.get('SQ_LDS_IDX_ACTIVE', 1)):5.2f} | {fmb:8.1f} {wmb:8.1f}")
lines.append('')
lines.append('per kernel (this forward): calls total_us  mfma_busy%(busy CUs)  fetchMB writeMB')
for nm, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"  {nm:36s} {a[0]:3d} {a[1]:8.1f}  {100 * a[2] / (4 * a[3]) if a[3] else float('nan'):6.1f}   {a[4]:8.1f} {a[5]:8.1f}")
mb_all = sum(a[2] for a in agg.values()); cu_all = sum(a[3] for a in agg.values())
GB = 1024 * 1024 / 1e9      # table columns are MiB (rocprofv3 reports KiB); totals in GB = 1e9 bytes
lines.append(f"TOTAL per forward ({frames} frames, {len(per)} launches): {tot_us:.1f} us kernel time; HBM fetch {tot_f * GB:.2f} GB "
             f"(FETCH_SIZE x calibrated factor: 2 for 16-B/lane loads, 1.25 stem_conv1, 1.40 conv3x3_rows) + write {tot_w * GB:.2f} GB = {(tot_f + tot_w) * GB:.2f} GB; "
             f"MFMA busy {100 * mb_all / (4 * cu_all) if cu_all else float('nan'):.1f} % of the SIMD-cycles of busy CUs")
open(out + '_per_kernel.txt', 'w').write('\n'.join(lines) + '\n')
rec = {"plan_hash": plan_hash, "frames_per_launch": frames, "launches": len(per),
       "hbm_bytes_per_launch": (tot_f + tot_w) * 1024 * 1024, "fetch_bytes_calibrated": tot_f * 1024 * 1024,
       "write_bytes": tot_w * 1024 * 1024, "kernel_time_us": tot_us, "algorithmic_mb_per_frame": alg_mb,
       "fetch_factor": "2 (16-B/lane loads; 1.94 measured on conv_igemm8), 1.25 stem_conv1, 1.40 conv3x3_rows",
       "fetch_calibration": "profiles/r03_fetch_calibration.json (exact bytes of single launches past the Infinity Cache / FETCH_SIZE)",
       "traffic_kind": "L2-miss (fabric-side) request bytes, Infinity-Cache hits included",
       "mfma_busy_frac_of_busy_cus": (mb_all / (4 * cu_all)) if cu_all else None,
       "source": "tools/pmc_collect.sh + tools/pmc_summary.py (rocprofv3 --pmc passes, counters only)"}
json.dump(rec, open(out + '_hbm_traffic.json', 'w'), indent=1)
print('\n'.join(lines[-(len(agg) + 3):]))
