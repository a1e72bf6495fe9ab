"""Per-phase breakdown inside trunk layer 2 of the ping-pong forward kernel (dev tool; needs a
-DNK_STAMPS=2 build in tools/ab/lib_stamps.so and NK_WAVES=8).  Stamps per stage: start, MFMAs
issued, after wait + barrier, after the stream-load issue, after the convert, after the stores;
the next stage's start closes the second barrier."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from nerf_kinematics_amd import lib as L_
L_.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", "lib_stamps.so")
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
names = ["mfma-issue", "wait+bar", "dma-issue", "convert", "stores", "bar2"]
for name, a in (("inference", None), ("save", act.data_ptr())):
    for _ in range(2):
        stamps.zero_()
        L_.check(lib.nk_mlp_forward(model._c, model.packed.data_ptr(), rb.data_ptr(), z.data_ptr(), R, n, raw.data_ptr(), a, st))
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(8, 512)
    for w in (0, 1, 4, 5):
        v = s[w]; k = int((v != 0).sum())
        v = v[:k].astype(np.float64)
        f = v[:48].reshape(8, 6)
        nxt = np.append(f[1:, 0], v[48])
        d = [np.median(f[:, i + 1] - f[:, i]) for i in range(5)] + [np.median(nxt - f[:, 5])]
        print(f"{name:9s} wave {w}: " + "  ".join(f"{nm} {x:5.0f}" for nm, x in zip(names, d)) + f"   stage {np.median(nxt - f[:, 0]):5.0f}  (layer {v[48] - v[0]:6.0f})")
    # absolute phase alignment of waves 0 and 4 (stage starts)
    print("  starts w0:", (s[0][:48:6] - s[0][0]).tolist())
    print("  starts w4:", (s[4][:48:6] - s[0][0]).tolist())
