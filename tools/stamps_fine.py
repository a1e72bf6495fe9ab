"""Per-stage breakdown inside trunk layer 2 of the forward chain kernel (dev tool; needs the
-DNK_STAMPS=2 build in tools/exp/lib_stamps.so).  Stamps per stage: start, after the
stream-load issue, after the tile stores, after the MFMA chain is issued, after wait + barrier;
the next stage's start closes the epilogue."""
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
for name, a in (("inference", None), ("save", act.data_ptr())):
    for _ in range(2):
        stamps.zero_()
        L_.check(lib.nk_mlp_forward(model._c, model.packed.data_ptr(), rb.data_ptr(), z.data_ptr(), R, n, raw.data_ptr(), a, st))
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(8, 512)
    for w in range(4):
        v = s[w]; k = int((v != 0).sum())
        v = v[:k].astype(np.float64)
        # section stamps: 0 start, 1 before trunk0, 2 after trunk0, 3 after L=1; then 8 x 5 fine stamps of L=2; then after L=2 ...
        f = v[4:4 + 40].reshape(8, 5)
        nxt = np.append(f[1:, 0], v[44])
        print(f"{name:9s} wave {w}: issue {np.median(f[:,1]-f[:,0]):5.0f}  stores {np.median(f[:,2]-f[:,1]):5.0f}  mfma {np.median(f[:,3]-f[:,2]):5.0f}  "
              f"wait+barrier {np.median(f[:,4]-f[:,3]):5.0f}  epilogue {np.median(nxt-f[:,4]):5.0f}   stage {np.median(nxt-f[:,0]):5.0f}   (layer {v[44]-v[3]:6.0f})")
