"""Time Trainer.step on fixed synthetic rays (dev tool).
    python tools/step_time.py [--rays N] [--small] [--lib path/to/variant.so]
--small: the 4x128 networks with 64+64 samples of the reference's lego experiment, else 8x256 with 64+128."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=16384); ap.add_argument("--small", action="store_true")
ap.add_argument("--lib"); ap.add_argument("--steps", type=int, default=30); ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
from nerf_kinematics_amd import lib as _lib
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
import torch
import nerf_kinematics_amd as nk
from nerf_kinematics_amd.trainer import Trainer, TrainOpts

R = a.rays
dev = torch.device("cuda:0")
arch = nk.ARCH_LEGO_CKPT if a.small else nk.ARCH_RUN_NERF
nf = 64 if a.small else 128
opts = TrainOpts(num_coarse=64, num_fine=nf, num_random_rays=R, perturb=True, radiance_field_noise_std=0.2, lr=5e-4, seed=3)
tr = Trainer(arch, arch, opts, dev)
g = torch.Generator().manual_seed(1)
o = torch.tensor([0.0, 0.0, 4.0]).expand(R, 3)
d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) * 0.2 + torch.tensor([0, 0, -1.0]), dim=-1)
rb = torch.cat([o, d, torch.full((R, 1), 2.0), torch.full((R, 1), 6.0), d], 1).contiguous().to(dev)
target = torch.rand(R, 3, generator=g).to(dev)
for _ in range(10):
    tr.step(rb, target)
for rep in range(a.reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.step(rb, target)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"{a.lib or 'libnerfkin.so'} {'4x128' if a.small else '8x256'} {R} rays: {dt * 1e3:.3f} ms/step, "
          f"{R * (64 + nf) / dt / 1e6:.1f} M ray-samples/s, loss {float(tr.loss):.6f}", flush=True)
