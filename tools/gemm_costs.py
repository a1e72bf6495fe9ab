"""Stage costs of the weight-gradient products, shape by shape (dev tool): what the ns-per-stage
column of NK_GEMM_SHAPES (csrc/nk_mlp_bwd.hip) is measured with.

    cd /tmp && rocprofv3 --kernel-trace -d <dir> -o t --output-format csv -- \
        python3 <repo>/bench.py --no-extras --no-cpu-baseline --steps 4 --warmup 1 --act-format <fmt>   (with NK_GEMM_PER_SHAPE=1)
    python tools/gemm_costs.py <dir>/t_kernel_trace.csv <fmt> [rays]

With NK_GEMM_PER_SHAPE=1 every k_dw_gemm launch holds the products of ONE tile shape, in the order
rgb, views, feature + alpha, plain trunk layers (6), skip layer, first layer -- fine network first,
then coarse.  Stage = 32 points (64 in the all-8-bit format); a launch is one round of <= 256
workgroups, so ns per stage = launch duration / stages per workgroup."""
import csv, sys

path, fmt = sys.argv[1], sys.argv[2]
rays = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
rows = [r for r in csv.DictReader(open(path)) if "k_dw_gemm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
shapes = [("rgb", 2, 1), ("views", 5, 1), ("feat+alpha", 3, 1), ("trunk", 0, 6), ("skip", 4, 1), ("trunk0", 1, 1)]
per_step = 2 * len(shapes)
assert len(rows) % per_step == 0, (len(rows), per_step)
steps = len(rows) // per_step
pts_per_stage = 64 if fmt == "fp8bf8" else 32
acc = {}
for s in range(1, steps):          # skip the first (warm-up) step
    for net, n in ((0, 192), (1, 64)):
        groups = (rays * n + 255) // 256 * 8
        for k, (name, sid, count) in enumerate(shapes):
            r = rows[s * per_step + net * len(shapes) + k]
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            stages = groups * 32 // pts_per_stage * count
            wgs = min(256, int(r.get("Grid_Size_X", r.get("Grid_Size", 256 * 512))) // 512) or 256
            acc.setdefault((net, name, sid), []).append(dur / (stages / wgs))
for (net, name, sid), v in sorted(acc.items()):
    print(f"{'fine' if net == 0 else 'coarse'} {name:11s} shape {sid}: {sum(v) / len(v):7.0f} ns per stage  ({len(v)} launches)")
