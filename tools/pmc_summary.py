"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (dev tool)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
t = {}
for r in rows:
    k = (r['Kernel_Name'][:44], r['Dispatch_Id'])
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    t[k] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
seen = collections.Counter()
for k, s in agg.items():
    if not ('k_mlp' in k[0] or 'gemm' in k[0]): continue
    seen[k[0]] += 1
    if seen[k[0]] > 2: continue
    wc = s.get('SQ_WAVE_CYCLES', 1)
    gui = s.get('GRBM_GUI_ACTIVE', 0) / 8
    mf = s.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
    print(f"{k[0]:46s} {t[k]:7.3f} ms  clk {gui / max(t[k], 1e-9) / 1e6:5.2f} GHz  mfma_util {mf / max(gui * 1024, 1):5.3f}  "
          f"wait_any {s.get('SQ_WAIT_ANY', 0) / wc:4.2f} wait_inst {s.get('SQ_WAIT_INST_ANY', 0) / wc:4.2f} active {s.get('SQ_ACTIVE_INST_ANY', 0) / wc:4.2f} "
          f"lds_wait {s.get('SQ_WAIT_INST_LDS', 0) / wc:5.3f}")
