"""Per-stage cycle breakdown of the forward chain kernel (dev tool; needs the NK_STAMPS build:
hipcc ... -DNK_STAMPS into tools/bin/libnerfkin_stamps.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from nerf_kinematics_amd import lib as L_
L_.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libnerfkin_stamps.so")
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
        v = s[w]; k = int((v != 0).sum()) // 4 * 4
        v = v[:k].reshape(-1, 4).astype(np.float64)
        d_issue = v[:, 1] - v[:, 0]; d_mfma = v[:, 2] - v[:, 1]; d_wait = v[:, 3] - v[:, 2]
        d_epi = v[1:, 0] - v[:-1, 3]
        tot = (v[-1, 3] - v[0, 0]) / (len(v) - 1)
        print(f"{name:10s} wave {w}: stages {len(v)}  per-stage {tot:7.0f}  issue+stores {d_issue.mean():6.0f}  mfma-issue {d_mfma.mean():6.0f}  "
              f"wait+barrier {d_wait.mean():6.0f}  epilogue(+layer gaps) {np.median(d_epi):6.0f} (mean {d_epi.mean():6.0f})")
    v = s[0]; k = int((v != 0).sum()) // 4 * 4
    print("  wave0 first 10 stages:", (v[:40].reshape(-1, 4)[:, [1, 2, 3]] - v[:40].reshape(-1, 4)[:, [0, 1, 2]]).tolist())
