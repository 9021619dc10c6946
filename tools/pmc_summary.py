"""Join rocprofv3 PMC passes (counter_collection.csv) per dispatch of the LAST forward and print per-kernel rows."""
import csv, sys, collections, os
root = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 50
per = collections.OrderedDict()
for p in sorted(os.listdir(root)):
    f = os.path.join(root, p, 'p_counter_collection.csv')
    if not os.path.exists(f): continue
    rows = list(csv.DictReader(open(f)))
    disp = collections.OrderedDict()
    for r in rows:
        d = int(r['Dispatch_Id'])
        disp.setdefault(d, {'name': r['Kernel_Name'], 'grid': int(r['Grid_Size']) // max(1, int(r['Workgroup_Size']))})
        disp[d][r['Counter_Name']] = float(r['Counter_Value'])
    ids = [d for d in disp if any(k in disp[d]['name'] for k in ('conv_igemm', 'stem', 'avgpool', 'conv1x1_pair', 'conv1x1_regw', 'conv3x3_narrow', 'conv3x3_rows'))]
    ids = ids[-n_last:]
    for i, d in enumerate(ids):
        per.setdefault(i, {}).update(disp[d])
cols = ['SQ_WAVE_CYCLES','SQ_BUSY_CYCLES','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_ACTIVE_INST_VALU','SQ_VALU_MFMA_BUSY_CYCLES','SQ_WAIT_INST_LDS','SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE','SQ_ACTIVE_INST_LDS','SQ_ACTIVE_INST_VMEM','SQ_INSTS_VALU','SQ_INSTS_MFMA','SQ_INSTS_LDS','SQ_INSTS_VMEM_RD','FETCH_SIZE','WRITE_SIZE','GRBM_GUI_ACTIVE']
print('idx kernel grid | wait_any% wait_inst% active% valu% mfma_busy%(of busy*4simd) lds_wait% | bankconf/lds_active  valu/mfma insts | fetchMB(x2) writeMB | gui_active')
for i, r in per.items():
    wc = r.get('SQ_WAVE_CYCLES', 1) or 1
    nm = r['name'].replace('void ', '').replace('(anonymous namespace)::', '')
    nm = nm.split('conv_igemm_kernel')[-1][:26] if 'conv_igemm' in nm else nm[:26]
    busy = r.get('SQ_BUSY_CYCLES', 1) or 1
    print(f"{i:2d} {nm:28s} {r['grid']:6d} | {100*r.get('SQ_WAIT_ANY',0)/wc:5.1f} {100*r.get('SQ_WAIT_INST_ANY',0)/wc:5.1f} {100*r.get('SQ_ACTIVE_INST_ANY',0)/wc:5.1f} {100*r.get('SQ_ACTIVE_INST_VALU',0)/wc:5.1f}  mfma_busy={r.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.3g} busy={busy:.3g} ldsw {100*r.get('SQ_WAIT_INST_LDS',0)/wc:5.1f} | {r.get('SQ_LDS_BANK_CONFLICT',0)/max(1,r.get('SQ_LDS_IDX_ACTIVE',1)):5.2f} {r.get('SQ_INSTS_VALU',0)/max(1,r.get('SQ_INSTS_MFMA',1)):6.1f} lds/mfma {r.get('SQ_INSTS_LDS',0)/max(1,r.get('SQ_INSTS_MFMA',1)):5.1f} | {2*r.get('FETCH_SIZE',0)/1024:8.1f} {r.get('WRITE_SIZE',0)/1024:8.1f} | {r.get('GRBM_GUI_ACTIVE',0):.3g}")

tot_f = sum(2 * r.get('FETCH_SIZE', 0) for r in per.values()) / 1024 / 1024
tot_w = sum(r.get('WRITE_SIZE', 0) for r in per.values()) / 1024 / 1024
print(f"TOTAL per forward: fetch {tot_f:.2f} GB (FETCH_SIZE x2 gfx950 correction)  write {tot_w:.2f} GB  sum {tot_f + tot_w:.2f} GB")
