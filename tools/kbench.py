"""Kernel micro-benchmarks: times single C-ABI stage calls with HIP events (dev tool)."""
import sys, os, math, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_kinematics_amd import lib as L_
if os.environ.get("NK_LIB"):            # experiment builds (tools/ab/*.so)
    L_.LIB_PATH = os.path.abspath(os.environ["NK_LIB"])
import nerf_kinematics_amd as nk

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=16384)
ap.add_argument("--n", type=int, default=192)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
arch = nk.ARCH_RUN_NERF
from nerf_kinematics_amd.trainer import _init_flat
model = nk.NeRFModel(arch, _init_flat(arch, 1).to(dev))
R, n = a.rays, a.n
rb = torch.randn(R, 11, device=dev)
z = torch.sort(torch.rand(R, n, device=dev) * 4 + 2, -1)[0]
lib = nk.load()
st = torch.cuda.current_stream().cuda_stream
raw = torch.empty(R, n, 4, device=dev)
act = nk.alloc_activations(model, R, n)
dz = torch.empty(L_.check(lib.nk_dz_bytes(model._c, R, n)), dtype=torch.uint8, device=dev)
grad = torch.zeros_like(model.flat)
d_raw = torch.randn(R, n, 4, device=dev)
FL = R * n * 1186816

def timeit(name, fn, flops):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"{name:28s} {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s")

timeit("mlp_forward (inference)", lambda: L_.check(lib.nk_mlp_forward(model._c, model.packed.data_ptr(), rb.data_ptr(), z.data_ptr(), R, n, raw.data_ptr(), None, st)), FL)
timeit("mlp_forward (save act)", lambda: L_.check(lib.nk_mlp_forward(model._c, model.packed.data_ptr(), rb.data_ptr(), z.data_ptr(), R, n, raw.data_ptr(), act.data_ptr(), st)), FL)
timeit("mlp_backward (chain+gemm)", lambda: L_.check(lib.nk_mlp_backward(model._c, model.packed.data_ptr(), act.data_ptr(), d_raw.data_ptr(), R, n, dz.data_ptr(), grad.data_ptr(), st)), 2 * FL)
