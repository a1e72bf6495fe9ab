"""Per-kernel statistics from a rocprofv3 rocpd database (dev tool): calls, total / mean duration,
share of the summed kernel time, and the gaps between consecutive kernels of the steady state.
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--csv out.csv] [--skip-first N]"""
import argparse, re, sqlite3, sys

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--csv")
ap.add_argument("--last", type=int, default=0, help="only the last N dispatches (steady state)")
a = ap.parse_args()
db = sqlite3.connect(a.db)
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, start, end from kernels order by start").fetchall()
if a.last:
    rows = rows[-a.last:]
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")
agg = {}
for n, s, e in rows:
    d = agg.setdefault(short(n), [0, 0])
    d[0] += 1; d[1] += e - s
tot = sum(v[1] for v in agg.values())
span = rows[-1][2] - rows[0][1]
gaps = sum(max(0, rows[i + 1][1] - rows[i][2]) for i in range(len(rows) - 1))
lines = ["kernel,calls,total_ms,mean_us,percent"]
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"\"{k}\",{c},{t / 1e6:.3f},{t / c / 1e3:.2f},{100 * t / tot:.2f}")
lines.append(f"# dispatches {len(rows)}, kernel time {tot / 1e6:.3f} ms, span {span / 1e6:.3f} ms, idle gaps {gaps / 1e6:.3f} ms ({100 * gaps / span:.1f} % of span)")
out = "\n".join(lines)
print(out)
if a.csv:
    open(a.csv, "w").write(out + "\n")
