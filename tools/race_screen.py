"""Race screen for the hand-synchronised kernels (dev tool): repeats the forward (must be
bit-identical run to run) and the backward (fp32 atomics: must agree to 1e-5 relative L2) many
times at several sizes and reports the worst deviation from the first run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_kinematics_amd as nk
from nerf_kinematics_amd.trainer import _init_flat

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for arch, name in ((nk.ARCH_RUN_NERF, "8x256"), (nk.ARCH_LEGO_CKPT, "4x128")):
    model = nk.NeRFModel(arch, _init_flat(arch, 1).to(dev))
    for R, n in ((16384, 192), (4096, 192), (1024, 64), (1000, 64), (37, 33)):
        g = torch.Generator(device="cpu").manual_seed(R + n)
        rb = torch.randn(R, 11, generator=g).to(dev)
        z = torch.sort(torch.rand(R, n, generator=g) * 4 + 2, -1)[0].to(dev)
        d_raw = torch.randn(R, n, 4, generator=g).to(dev)
        act = nk.alloc_activations(model, R, n)
        raw0 = nk.run_network(model, rb, z, act_save=act).clone()
        act0 = act.clone()
        g0 = nk.mlp_backward(model, act, d_raw).clone()
        worst_f, worst_a, worst_g = 0, 0, 0.0
        for _ in range(reps):
            raw = nk.run_network(model, rb, z, act_save=act)
            worst_f += int(not torch.equal(raw, raw0))
            worst_a += int(not torch.equal(act, act0))
            gi = nk.mlp_backward(model, act, d_raw)
            worst_g = max(worst_g, float((gi - g0).norm() / g0.norm()))
        torch.cuda.synchronize()
        ok = worst_f == 0 and worst_a == 0 and worst_g < 1e-5
        bad += not ok
        print(f"{name} R={R:6d} n={n:3d}: forward mismatches {worst_f}/{reps}, saved-activation mismatches {worst_a}/{reps}, "
              f"worst gradient deviation {worst_g:.2e}  {'ok' if ok else 'FAIL'}", flush=True)
sys.exit(1 if bad else 0)
