import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nerf_kinematics_amd as nk
from nerf_kinematics_amd import lib as L_
from oracle import nerf_oracle as O
dev = torch.device("cuda:0")
arch = nk.ARCH_LEGO_CKPT
oa = O.ARCH_LEGO_CKPT
flat = O.NeRFNet.init_flat(oa, 7)
R, n = 20, 64
g = torch.Generator().manual_seed(11)
focal = 0.5 * 16 / math.tan(0.5 * 0.6911112070083618)
ro, rd = O.get_rays(16, 16, focal, O.spherical_pose(45.0, -30.0, 4.0))
rb = O.make_ray_batch(ro, rd, 2.0, 6.0)[torch.randperm(256, generator=g)[:R]].contiguous()
z = torch.sort(torch.rand(R, n, generator=g) * 4 + 2, -1)[0]
d_raw = torch.randn(R, n, 4, generator=g)
model = nk.NeRFModel(arch, flat.to(dev))
act = nk.alloc_activations(model, R, n)
raw = nk.run_network(model, rb.to(dev), z.to(dev), act_save=act)
lib = nk.load()
dzb = L_.check(lib.nk_dz_bytes(model._c, R, n))
dz = torch.zeros(dzb, dtype=torch.uint8, device=dev)
grad = torch.zeros_like(model.flat)
dr = d_raw.to(dev).contiguous()
L_.check(lib.nk_mlp_backward(model._c, model.packed.data_ptr(), act.data_ptr(), dr.data_ptr(), R, n, dz.data_ptr(), grad.data_ptr(), torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
P = R * n
G = ((P + 255) // 256) * 8
# region 0: dOut [G][4][32][8] bf16
reg0 = dz[: G * 4 * 32 * 8 * 2].view(torch.bfloat16).view(G, 4, 32, 8).float().cpu()
dout = reg0.permute(0, 2, 1, 3).reshape(G * 32, 32)[:P]   # [P, 32 features]
print("dOut feature sums (first 6):", dout.sum(0)[:6].tolist())
ref = d_raw.reshape(-1, 4).to(torch.bfloat16).float().sum(0)
print("bf16(d_raw) sums:", ref.tolist())
# offsets in flat for flexible: find via arch.param_shapes
off = 0; offs = {}
for name, (o, i) in arch.param_shapes():
    offs[name + ".weight"] = (off, o * i); off += o * i
    offs[name + ".bias"] = (off, o); off += o
gc = grad.cpu()
for k in ("fc_rgb.bias", "fc_alpha.bias"):
    a, c = offs[k]
    print(k, gc[a:a + c].tolist())
# weight check for alpha: sum_p dsigma * h
