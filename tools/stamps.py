"""Per-section time breakdown of the forward chain kernel (dev tool; needs the NK_STAMPS build:
every source compiled with -DNK_STAMPS into tools/exp/lib_stamps.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from nerf_kinematics_amd import lib as L_
L_.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "lib_stamps.so")
import nerf_kinematics_amd as nk
from nerf_kinematics_amd.trainer import _init_flat

dev = torch.device("cuda:0")
arch = nk.ARCH_RUN_NERF
model = nk.NeRFModel(arch, _init_flat(arch, 1).to(dev))
R, n = 16384, 192
rb = torch.randn(R, 11, device=dev)
z = torch.sort(torch.rand(R, n, device=dev) * 4 + 2, -1)[0]
lib = nk.load()
st = torch.cuda.current_stream().cuda_stream
raw = torch.empty(R, n, 4, device=dev)
act = nk.alloc_activations(model, R, n)
stamps = torch.zeros(8 * 512, dtype=torch.int64, device=dev)
os.environ["NK_STAMP_PTR"] = str(stamps.data_ptr())
names = ["prologue(PE, first loads)", "trunk0"] + [f"trunk{L}" for L in range(1, 8)] + ["alpha", "feat", "views", "rgb+store"]
for name, a in (("inference", None), ("save", act.data_ptr())):
    for _ in range(2):
        stamps.zero_()
        L_.check(lib.nk_mlp_forward(model._c, model.packed.data_ptr(), rb.data_ptr(), z.data_ptr(), R, n, raw.data_ptr(), a, st))
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(8, 512)
    v = s[0]; k = int((v != 0).sum())
    d = np.diff(v[:k].astype(np.float64))
    tot = d.sum()
    print(f"{name}: {k} stamps, total {tot:.0f} units")
    for nm, x in zip(names, d):
        print(f"   {nm:28s} {x:8.0f}  {100 * x / tot:5.1f}%")
