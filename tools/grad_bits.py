"""Checksums of the render outputs and gradients of a few fixed batches (dev tool): run it with two
builds of the library (NK_LIB=<other .so>) to see whether a change kept the default path's bits."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_kinematics_amd as nk
from nerf_kinematics_amd.datasets import RaySampler, teacher_scene
from nerf_kinematics_amd.trainer import _init_flat

dev = torch.device("cuda:0")
scene = teacher_scene(None, 4, 0, 100, dev)
sha = lambda t: hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]
for name, arch, R in (("8x256", nk.ARCH_RUN_NERF, 6001), ("8x256", nk.ARCH_RUN_NERF, 1024), ("4x128", nk.ARCH_LEGO_CKPT, 16384),
                      ("4x128", nk.ARCH_LEGO_CKPT, 777)):
    rb, tgt = RaySampler(scene, R, seed=5).next(0)
    c, f = nk.NeRFModel(arch, _init_flat(arch, 1).to(dev)), nk.NeRFModel(arch, _init_flat(arch, 2).to(dev))
    c.flat.requires_grad_(True); f.flat.requires_grad_(True)
    out = nk.render_rays(rb, c, f, num_coarse=64, num_fine=128, perturb=True, radiance_field_noise_std=0.2, seed=3)
    loss = ((out[3] - tgt) ** 2).mean() + ((out[0] - tgt) ** 2).mean()
    loss.backward()
    with torch.no_grad():
        inf = nk.render_rays(rb, c, f, num_coarse=64, num_fine=128, training=False)
    print(name, R, "rgb", sha(out[0]), "grad_c", sha(c.flat.grad), "grad_f", sha(f.flat.grad), "inference", sha(inf[0]))
