"""Per-kernel table of one bench run (dev tool): mean launch duration from the rocprofv3 kernel
trace, HBM bytes per launch from the two PMC passes (WRITE_SIZE + 2 x FETCH_SIZE: on gfx950
FETCH_SIZE tallies 128-byte read requests at 64 bytes), bandwidth = bytes / duration, matrix-pipe
utilisation and clock from the SQ pass (SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs x GRBM_GUI_ACTIVE
/ 8 XCDs).
    python tools/kernel_table.py <kernel_stats.csv> <fetch counter_collection.csv> <write ...> <sq ...> > out.csv"""
import collections, csv, re, sys

stats, fetch, write, sq = sys.argv[1:5]
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "").strip()


def per_kernel(path):
    """kernel -> counter -> mean value per dispatch, and kernel -> mean duration (ms) in that run"""
    val = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        val[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
        dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return ({k: {c: v / len(disp[k]) for c, v in d.items()} for k, d in val.items()},
            {k: sum(d.values()) / len(d) for k, d in dur.items()})


f, _ = per_kernel(fetch)
w, _ = per_kernel(write)
s, sdur = per_kernel(sq)
print("kernel,calls,mean_ms,percent_of_kernel_time,hbm_read_MB_per_launch,hbm_write_MB_per_launch,TB_per_s,mfma_util,clock_GHz,wait_any,active")
for r in csv.DictReader(open(stats)):
    k = short(r["Name"])
    ms = float(r["AverageNs"]) / 1e6
    rd = 2 * f.get(k, {}).get("FETCH_SIZE", 0.0) * 1024 / 1e6
    wr = w.get(k, {}).get("WRITE_SIZE", 0.0) * 1024 / 1e6
    c = s.get(k, {})
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8
    util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024) if gui else 0.0
    clk = gui / (sdur.get(k, 0) * 1e6) if sdur.get(k) else 0.0
    wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    print(f"\"{k}\",{r['Calls']},{ms:.4f},{r['Percentage']},{rd:.1f},{wr:.1f},{(rd + wr) / 1e6 / (ms * 1e-3):.3f},"
          f"{util:.3f},{clk:.2f},{c.get('SQ_WAIT_ANY', 0) / wc:.2f},{c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f}")
