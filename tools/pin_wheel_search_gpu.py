"""Global search for the camera (rotation, translation, focal length) under which the reference's
trained wheel_robot6 networks reproduce the reference's own 160 x 90 validation render
(tests/golden/wheel_ref_render_160x90.png).  Dev tool, GPU: candidates are rendered by the HIP path
(NDC rays, 64 + 128 samples), thousands per second, scored by correlation with the reference
render; the winners are to be re-checked with the CPU oracle (tools/pin_wheel_search.py scoring).

Stage A: random cameras from a broad prior (uniform rotations, translation 1e-3 .. 2e3 in any
direction, focal 21 .. 670 px) at every second pixel.  Stage B: Gaussian hill-climbing around the
best of stage A at full resolution.
    python tools/pin_wheel_search_gpu.py [--random N] [--seconds S] > gpurun_out/wheel_pin_gpu.txt"""
import argparse, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image
from scipy.spatial.transform import Rotation as Rot
import nerf_kinematics_amd as nk

ap = argparse.ArgumentParser()
ap.add_argument("--random", type=int, default=200000)
ap.add_argument("--seconds", type=float, default=420.0)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
G = os.path.join(ROOT, "tests", "golden")
dev = torch.device("cuda:0")
w = np.load(os.path.join(G, "wheel_robot6_weights.npz"))
coarse = nk.NeRFModel(nk.ARCH_WHEEL_CKPT, torch.from_numpy(w["coarse"]).to(dev))
fine = nk.NeRFModel(nk.ARCH_WHEEL_CKPT, torch.from_numpy(w["fine"]).to(dev))
gen = torch.from_numpy(np.asarray(Image.open(os.path.join(G, "wheel_ref_render_160x90.png"))).astype(np.float32) / 255).to(dev)
H, W = 90, 160
rng = np.random.default_rng(a.seed)


def pixel_set(stride):
    jj, ii = np.meshgrid(np.arange(stride // 2, H, stride), np.arange(stride // 2, W, stride), indexing="ij")
    pix = torch.from_numpy((jj * W + ii).reshape(-1).astype(np.int32)).to(dev)
    return pix, gen.reshape(-1, 3)[pix.long()]


def score(cands, pix, tgt):
    """cands: list of (R 3x3, t 3, focal) -> correlation of each render with the reference render"""
    n = pix.numel()
    rays = torch.empty(len(cands) * n, 11, device=dev)
    for k, (R, t, f) in enumerate(cands):
        c2w = np.concatenate([R, t[:, None]], 1)
        rays[k * n:(k + 1) * n] = nk.get_rays(H, W, float(f), c2w, near=0.0, far=1.0, ndc=True, pixels=pix)
    with torch.no_grad():
        rgb = nk.render_rays(rays, coarse, fine, num_coarse=64, num_fine=128)[0]
    rgb = torch.nan_to_num(rgb.reshape(len(cands), n * 3), nan=0.0, posinf=0.0, neginf=0.0)
    t_ = tgt.reshape(1, -1)
    x = rgb - rgb.mean(1, keepdim=True); y = t_ - t_.mean()
    corr = (x * y).sum(1) / (x.norm(dim=1) * y.norm() + 1e-12)
    mse = ((rgb - t_) ** 2).mean(1)
    return corr.cpu().numpy(), (-10 * torch.log10(mse)).cpu().numpy()


def random_cand():
    R = Rot.random(random_state=rng.integers(1 << 31)).as_matrix()
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    return R, d * 10 ** rng.uniform(-3, 3.3), 84.3 * 2 ** rng.uniform(-2, 3)


t0 = time.time()
pix2, tgt2 = pixel_set(2)
pool = []
B = 48
done = 0
while done < a.random and time.time() - t0 < a.seconds * 0.5:
    cands = [random_cand() for _ in range(B)]
    c, p = score(cands, pix2, tgt2)
    pool += [(float(c[i]), float(p[i]), cands[i]) for i in range(B) if np.isfinite(c[i])]
    done += B
    if len(pool) > 4000:
        pool = sorted(pool, key=lambda r: -r[0])[:400]
pool = sorted(pool, key=lambda r: -r[0])[:400]
print(f"# stage A: {done} random cameras in {time.time() - t0:.0f} s; best correlations: " + " ".join(f"{r[0]:.3f}" for r in pool[:10]), flush=True)

pix1, tgt1 = pixel_set(1)
best = []
t1 = time.time()
k = 0
while time.time() - t0 < a.seconds and k < len(pool):
    c0, _, (R, t, f) = pool[k]; k += 1
    rv = Rot.from_matrix(R).as_rotvec()
    x = np.concatenate([rv, t, [math.log(f)]])
    sig = np.concatenate([[0.15] * 3, np.maximum(np.abs(t) * 0.3, 1e-3), [0.2]])
    cur = -1.0
    for gen_i in range(24):
        pop = [x] + [x + sig * rng.normal(size=7) for _ in range(31)]
        cands = [(Rot.from_rotvec(q[:3]).as_matrix(), q[3:6], math.exp(q[6])) for q in pop]
        c, p = score(cands, pix1, tgt1)
        i = int(np.nanargmax(c))
        if c[i] > cur:
            cur, x, psnr = float(c[i]), pop[i], float(p[i])
        sig *= 0.85
    best.append((cur, psnr, x))
    print(f"stage B start {k} (A {c0:.3f}): corr {cur:.4f} psnr {psnr:.2f} rotvec {np.round(x[:3], 4).tolist()} t {np.round(x[3:6], 4).tolist()} f {math.exp(x[6]):.2f}", flush=True)
best.sort(key=lambda r: -r[0])
print("# best:")
for c, p, x in best[:5]:
    print(f"corr {c:.4f} psnr {p:.2f} rotvec {x[:3].tolist()} t {x[3:6].tolist()} focal {math.exp(x[6]):.4f}")
