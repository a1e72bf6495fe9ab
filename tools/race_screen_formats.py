"""Race screen at the render_rays level (dev tool): for every saved-tensor format and several batch
sizes, repeats forward + backward and the inference render (the 16x16x32 kernel) and counts runs
whose outputs or gradients differ in any bit from the first run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_kinematics_amd as nk
from nerf_kinematics_amd.datasets import RaySampler, teacher_scene
from nerf_kinematics_amd.trainer import _init_flat

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
scene = teacher_scene(None, 4, 0, 100, dev)
bad = 0
A = nk.Arch
shapes = ((nk.ARCH_RUN_NERF, "8x256"), (nk.ARCH_LEGO_CKPT, "4x128"),
          (A(8, 256, 4, 128, 10, 4, True, False, "run_nerf", False), "8x256 no viewdirs"),
          (A(4, 128, -1, 64, 10, 4, False, True, "flexible", False), "4x128 no viewdirs"))
if len(sys.argv) > 2:       # a substring selects shapes
    shapes = tuple(s for s in shapes if sys.argv[2] in s[1])
for arch, name in shapes:
    for R in (16384, 5507, 1024, 37):
        rb, tgt = RaySampler(scene, R, seed=5).next(0)
        for fmt in ("bf16", "fp8", "fp8bf8"):
            first = None
            mism = 0
            for rep in range(reps):
                c, f = nk.NeRFModel(arch, _init_flat(arch, 1).to(dev)), nk.NeRFModel(arch, _init_flat(arch, 2).to(dev))
                c.flat.requires_grad_(True); f.flat.requires_grad_(True)
                out = nk.render_rays(rb, c, f, num_coarse=64, num_fine=128, perturb=True, radiance_field_noise_std=0.2, seed=3,
                                     act_format=fmt)
                (((out[3] - tgt) ** 2).mean() + ((out[0] - tgt) ** 2).mean()).backward()
                with torch.no_grad():
                    inf = nk.render_rays(rb, c, f, num_coarse=64, num_fine=128, skip_empty_eps=1e-3 if rep % 2 else 0.0)
                cur = (out[0].detach().clone(), c.flat.grad.clone(), f.flat.grad.clone(), inf[0].clone() if rep % 2 == 0 else None,
                       inf[0].clone() if rep % 2 else None)
                if first is None:
                    first = list(cur)
                    continue
                for k in range(5):
                    if cur[k] is None:
                        continue
                    if first[k] is None:
                        first[k] = cur[k]
                    elif not torch.equal(cur[k], first[k]):
                        mism += 1
            torch.cuda.synchronize()
            bad += mism
            print(f"{name} R={R:6d} {fmt:6s}: {mism} mismatching tensors over {reps} repeats (rgb, both gradients, inference render, "
                  f"skipped render)  {'ok' if mism == 0 else 'FAIL'}", flush=True)
sys.exit(1 if bad else 0)
