"""Long training runs of the 8x256 pair under each saved-tensor format (dev tool): does an 8-bit
format cost held-out PSNR late in training, when gradients are small?  One seed per format, the same
rays and draws in every run; held-out PSNR of the fine map (validation settings) at a few iterations.

    python tools/long_run_formats.py [iterations] [rays per step] [formats...]     (SEEDS="42 43 ..." for more than one seed)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_kinematics_amd as nk
from nerf_kinematics_amd.datasets import RaySampler, teacher_scene
from nerf_kinematics_amd.metrics import mse2psnr
from nerf_kinematics_amd.trainer import Trainer, TrainOpts

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
formats = sys.argv[3:] or ["bf16", "fp8", "fp8bf8"]
dev = torch.device("cuda:0")
scene = teacher_scene(None, 40, 4, 200, dev)
marks = sorted({m for m in (500, 1000, 2000, 5000, 10000, 20000, 40000, iters) if m <= iters})
print(f"8x256 pair, 64 + 128 samples, {rays} rays per step, lr 5e-4 decaying 0.1 per 250 000, 40 teacher views of 200x200, "
      f"4 held-out; PSNR (dB) of the fine map at iterations {marks}", flush=True)
for seed, fmt in ((sd, f) for sd in (int(x) for x in os.environ.get("SEEDS", "42").split()) for f in formats):
    print(f"seed {seed}", flush=True)
    tr = Trainer(nk.ARCH_RUN_NERF, nk.ARCH_RUN_NERF, TrainOpts(num_coarse=64, num_fine=128, num_random_rays=rays, perturb=True,
                 radiance_field_noise_std=0.2, lr=5e-4, seed=seed, act_format=fmt), dev, rays_per_step=rays, init_seed=seed)
    smp = RaySampler(scene, rays, seed=seed)
    row, t0, train_s = [], time.perf_counter(), 0.0
    for i in range(iters):
        loss = tr.step(*smp.next(i))
        if i + 1 in marks:
            torch.cuda.synchronize()
            train_s += time.perf_counter() - t0
            vals = []
            for k in scene.i_val:
                out = tr.render(scene.rays(int(k), dev))
                vals.append(mse2psnr(float(((out["rgb_fine"] - scene.target(int(k))) ** 2).mean())))
            row.append(sum(vals) / len(vals))
            print(f"  {fmt:6s} iteration {i + 1:6d}: held-out {row[-1]:.2f} dB, training loss {float(loss):.5f}", flush=True)
            t0 = time.perf_counter()
    print(f"{fmt:6s} " + " ".join(f"{v:6.2f}" for v in row) + f"   ({train_s:.0f} s of training, {1e3 * train_s / iters:.2f} ms per step)", flush=True)
