"""Randomised parity sweep of render_rays against the oracle (dev tool): random network shapes (both
lineages, with and without view directions, depth, skip, encoding lengths and variants), sample counts, ray
counts, flags and saved-tensor formats; compares the rgb and opacity maps of a training call and the
gradient of both networks with the bf16-emulating oracle (which emulates the 8-bit formats too).

    python tools/fuzz_parity.py [cases] [seed]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import nerf_kinematics_amd as nk
from nerf_kinematics_amd import render as render_mod
from oracle import nerf_oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")


def oarch(a):
    return O.Arch(a.depth, a.width, a.skip, a.dir_width, a.n_freq_xyz, a.n_freq_dir, a.relu_first, a.relu_feat, a.lineage,
                  a.use_viewdirs, a.include_input_xyz, a.include_input_dir, a.log_sampling_xyz, a.log_sampling_dir)


def random_arch():
    W = int(rng.choice([128, 256]))
    lineage = str(rng.choice(["flexible", "run_nerf"]))
    D = int(rng.integers(2, 9))
    skip = -1 if lineage == "flexible" or D < 3 or rng.random() < 0.3 else int(rng.integers(0, D - 1))
    return nk.Arch(D, W, skip, W // 2, int(rng.integers(1, 11)), int(rng.integers(1, 5)), bool(rng.integers(0, 2)),
                   bool(rng.integers(0, 2)), lineage, use_viewdirs=bool(rng.random() < 0.7),
                   include_input_xyz=bool(rng.random() < 0.7), include_input_dir=bool(rng.random() < 0.7),
                   log_sampling_xyz=bool(rng.random() < 0.7), log_sampling_dir=bool(rng.random() < 0.7))


bad = 0
for case in range(cases):
    arch_c, arch_f = random_arch(), random_arch()
    if rng.random() < 0.5:
        arch_f = arch_c
    Nc = int(rng.integers(2, 97))
    Nf = int(rng.choice([0, int(rng.integers(1, 161))]))
    R = int(rng.choice([1, 2, 31, 33, int(rng.integers(3, 200))]))
    ndc = bool(rng.random() < 0.25)
    kw = dict(perturb=bool(rng.integers(0, 2)), lindisp=bool(rng.random() < 0.3) and not ndc,
              radiance_field_noise_std=float(rng.choice([0.0, 0.2, 1.0])), white_background=bool(rng.random() < 0.3),
              seed=int(rng.integers(0, 2 ** 31)))
    fmt = str(rng.choice(["bf16", "bf16", "fp8", "fp8bf8"]))
    H = Wd = 24
    focal = 0.5 * Wd / math.tan(0.5 * 0.6911112070083618)
    g = torch.Generator().manual_seed(case)
    if ndc:
        c2w = torch.eye(4)[:3]
        ro, rd = O.get_rays(H, Wd, focal, c2w)
        ro = ro + torch.tensor([0.1, -0.2, 0.0])
        ro, rd = O.ndc_rays(H, Wd, focal, 1.0, ro.reshape(-1, 3), rd.reshape(-1, 3))
        rb = O.make_ray_batch(ro, rd, 0.0, 1.0)
    else:
        ro, rd = O.get_rays(H, Wd, focal, O.spherical_pose(float(rng.uniform(0, 360)), float(rng.uniform(-80, -5)), 4.0))
        rb = O.make_ray_batch(ro, rd, 2.0, 6.0)
    rb = rb[torch.randperm(H * Wd, generator=g)[:R]].contiguous()
    need_dirs = arch_c.use_viewdirs or (Nf > 0 and arch_f.use_viewdirs)
    if not need_dirs and rng.random() < 0.5:
        rb = rb[:, :8].contiguous()
    fc = O.NeRFNet.init_flat(oarch(arch_c), case)
    ff = O.NeRFNet.init_flat(oarch(arch_f), case + 1000)
    # lift the density head's bias so that rays are not all transparent
    for arch, flat in ((arch_c, fc), (arch_f, ff)):
        off = 0
        for name, (o, i) in arch.param_shapes():
            off += o * i
            if name in ("fc_alpha", "alpha_linear"):
                flat[off] += 0.5
            elif name in ("fc_out", "output_linear"):
                flat[off + 3] += 0.5
            off += o
    target = torch.rand(R, 3, generator=g)
    tag = (f"case {case}: coarse {arch_c.depth}x{arch_c.width} {arch_c.lineage} skip {arch_c.skip} L {arch_c.n_freq_xyz}/{arch_c.n_freq_dir} "
           f"dirs {int(arch_c.use_viewdirs)} enc {arch_c.enc_flags} | fine {arch_f.depth}x{arch_f.width} {arch_f.lineage} skip {arch_f.skip} dirs {int(arch_f.use_viewdirs)} "
           f"| R {R} Nc {Nc} Nf {Nf} cols {rb.shape[1]} ndc {int(ndc)} {fmt} {kw}")
    try:
        c = nk.NeRFModel(arch_c, fc.to(dev)); f = nk.NeRFModel(arch_f, ff.to(dev)) if Nf > 0 else None
        c.flat.requires_grad_(True)
        if f is not None:
            f.flat.requires_grad_(True)
        out = nk.render_rays(rb.to(dev), c, f, num_coarse=Nc, num_fine=Nf, act_format=fmt, **kw)
        tg = target.to(dev)
        loss = ((out[3] - tg) ** 2).mean() + (((out[0] - tg) ** 2).mean() if Nf > 0 else 0.0)
        loss.backward()
        fco, ffo = fc.clone().requires_grad_(True), ff.clone().requires_grad_(True)
        S = render_mod.dz_scale(R)
        co = O.NeRFNet.from_flat(oarch(arch_c), fco, emulate_bf16=True, act_format=fmt, dz_scale=S)
        fo = O.NeRFNet.from_flat(oarch(arch_f), ffo, emulate_bf16=True, act_format=fmt, dz_scale=S) if Nf > 0 else None
        ref = O.render_rays(rb, co, fo, O.RenderOpts(Nc, Nf, **kw))
        lref = ((ref["rgb_coarse"] - target) ** 2).mean() + (((ref["rgb_fine"] - target) ** 2).mean() if Nf > 0 else 0.0)
        lref.backward()
        errs = {"rgb_c": float((out[3].detach().cpu() - ref["rgb_coarse"].detach()).abs().max()),
                "acc_c": float((out[5].detach().cpu() - ref["acc_coarse"].detach()).abs().max())}
        if Nf > 0:
            errs["rgb_f"] = float((out[0].detach().cpu() - ref["rgb_fine"].detach()).abs().max())
            errs["acc_f"] = float((out[2].detach().cpu() - ref["acc_fine"].detach()).abs().max())
        gc = float((c.flat.grad.cpu() - fco.grad).norm() / (fco.grad.norm() + 1e-20))
        gf = float((f.flat.grad.cpu() - ffo.grad).norm() / (ffo.grad.norm() + 1e-20)) if Nf > 0 else 0.0
        gnorm = float(fco.grad.norm())
        gtol = 5e-2          # whole-gradient relative L2 (the per-tensor bounds of the tests are 5e-2 / 1e-1)
        ok = max(errs.values()) < 2e-2 and (gc < gtol or gnorm < 1e-7) and (gf < gtol or float(ffo.grad.norm()) < 1e-7) \
            and torch.isfinite(c.flat.grad).all()
        print(("ok   " if ok else "FAIL ") + tag + f" -> max map err {max(errs.values()):.2e}, grad rel coarse {gc:.3e} fine {gf:.3e} (|g| {gnorm:.2e})", flush=True)
        bad += 0 if ok else 1
    except Exception as e:      # noqa: BLE001 (a sweep reports and goes on)
        print("ERR  " + tag + f" -> {type(e).__name__}: {e}", flush=True)
        bad += 1
print(f"{bad} of {cases} cases failed")
sys.exit(1 if bad else 0)
