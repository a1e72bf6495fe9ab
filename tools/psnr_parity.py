"""HIP training run against the fp32 CPU restatement on identical inputs (dev tool): same initial
parameters, same rays and colours every iteration, same jitter / noise seeds, same Adam and
learning-rate schedule; prints held-out PSNR of both at checkpoints."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import nerf_kinematics_amd as nk
from nerf_kinematics_amd import datasets
from nerf_kinematics_amd.metrics import mse2psnr
from nerf_kinematics_amd.trainer import Trainer, TrainOpts, _init_flat, lr_at
from oracle import nerf_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--rays", type=int, default=512)
ap.add_argument("--res", type=int, default=32)
ap.add_argument("--every", type=int, default=100)
ap.add_argument("--seed", type=int, default=42)
ap.add_argument("--lr", type=float, default=5e-3)
ap.add_argument("--act-format", default="bf16", choices=("bf16", "fp8", "fp8bf8"), help="saved-tensor format of the HIP run")
ap.add_argument("--no-viewdirs", action="store_true", help="networks without the direction branch (use_viewdirs: False)")
a = ap.parse_args()
dev = torch.device("cuda:0")
Nc, Nf = 32, 32
scene = datasets.teacher_scene(None, 12, 3, a.res, dev, num_coarse=64, num_fine=64)
arch = nk.ARCH_LEGO_CKPT
if a.no_viewdirs:
    arch = nk.Arch(arch.depth, arch.width, arch.skip, arch.dir_width, arch.n_freq_xyz, arch.n_freq_dir, arch.relu_first,
                   arch.relu_feat, arch.lineage, use_viewdirs=False)
oa = O.Arch(arch.depth, arch.width, arch.skip, arch.dir_width, arch.n_freq_xyz, arch.n_freq_dir, arch.relu_first,
            arch.relu_feat, arch.lineage, arch.use_viewdirs)
opts = TrainOpts(num_coarse=Nc, num_fine=Nf, num_random_rays=a.rays, perturb=True, radiance_field_noise_std=0.2,
                 lr=a.lr, seed=a.seed, act_format=a.act_format)
tr = Trainer(arch, arch, opts, dev)
smp = datasets.RaySampler(scene, a.rays, seed=a.seed)
fc = _init_flat(arch, a.seed).clone().requires_grad_(True)
ff = _init_flat(arch, a.seed + 1).clone().requires_grad_(True)
opt = torch.optim.Adam([fc, ff], lr=opts.lr, betas=opts.betas, eps=opts.eps)
val_rays = [scene.rays(int(k), dev) for k in scene.i_val]
val_tgt = [scene.target(int(k)) for k in scene.i_val]

def psnr_hip():
    v = []
    for rb, t in zip(val_rays, val_tgt):
        out = tr.render(rb, opts=dict(num_coarse=Nc, num_fine=Nf))
        v.append(mse2psnr(float(((out["rgb_fine"] - t) ** 2).mean())))
    return float(np.mean(v))

@torch.no_grad()
def psnr_oracle():
    co, fo = O.NeRFNet.from_flat(oa, fc), O.NeRFNet.from_flat(oa, ff)
    v = []
    for rb, t in zip(val_rays, val_tgt):
        out = O.render_rays(rb.cpu(), co, fo, O.RenderOpts(Nc, Nf, perturb=False, radiance_field_noise_std=0.0))
        v.append(mse2psnr(float(((out["rgb_fine"] - t.cpu()) ** 2).mean())))
    return float(np.mean(v))

t0 = time.time()
for i in range(a.iters):
    rb, tgt = smp.next(i)
    seed = (opts.seed + 0x9E3779B1 * (i + 1)) & 0xFFFFFFFF
    lh = float(tr.step(rb, tgt))
    co, fo = O.NeRFNet.from_flat(oa, fc), O.NeRFNet.from_flat(oa, ff)
    out = O.render_rays(rb.cpu(), co, fo, O.RenderOpts(Nc, Nf, perturb=True, radiance_field_noise_std=0.2, seed=seed))
    lo = O.nerf_loss(out, tgt.cpu())
    for g in opt.param_groups:
        g["lr"] = lr_at(i, opts)
    opt.zero_grad(); lo.backward(); opt.step()
    if (i + 1) % a.every == 0 or i == 0:
        print(f"iter {i + 1:5d}  loss hip {lh:.5f} oracle {float(lo.detach()):.5f}  held-out PSNR hip {psnr_hip():.3f} oracle {psnr_oracle():.3f}  "
              f"[{time.time() - t0:.0f} s]", flush=True)
