"""The two 1280 x 720 forward-facing configurations at FULL size on the GPU: the synthetic
forward-facing grid (NDC, ten position frequencies) and the reference's robot capture at
``downsample_factor: 1`` (NDC, six position frequencies, the reference's trained wheel weights).

At this size the CPU oracle checks a ray subset (same rays, same weights: image-level tolerance of
the bf16 MLP, mean |d rgb| < 3e-3 for the lego field, < 8e-3 for the wheel field); the whole frame is held to size-independent properties:
chunk invariance bit for bit, opacity within [0, 1], rgb within [0, 1], seed determinism.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nk():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU and the HIP library; there is no CPU fallback")
    import nerf_kinematics_amd as nk
    nk.load()
    return nk


def _oracle_rays(scene, k, sel):
    """Oracle ray rows of pixels `sel` of view k (pinhole rays, NDC warp when the scene asks)."""
    c2w = torch.from_numpy(scene.poses[k][:3, :4].astype(np.float32))
    ro, rd = O.get_rays(scene.H, scene.W, (scene.fx, scene.fy), c2w, scene.cx, scene.cy)
    ro, rd = ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]
    vd = rd / rd.norm(dim=-1, keepdim=True)
    if scene.ndc:
        ro, rd = O.ndc_rays(scene.H, scene.W, (scene.fx, scene.fy), 1.0, ro, rd)
    n = ro.shape[0]
    return torch.cat([ro, rd, torch.full((n, 1), float(scene.near)), torch.full((n, 1), float(scene.far)), vd], -1)


def _full_frame_properties(nk, scene, k, coarse, fine, dev):
    rb = scene.rays(k, dev)
    assert rb.shape == (1280 * 720, 11)
    kw = dict(num_coarse=64, num_fine=128)
    with torch.no_grad():
        a = [torch.cat([nk.render_rays(rb[s:s + c], coarse, fine, **kw)[i] for s in range(0, rb.shape[0], c)])
             for c in (131072, 100000) for i in (0, 2)]
    rgb_a, acc_a, rgb_b, acc_b = a
    assert torch.equal(rgb_a, rgb_b) and torch.equal(acc_a, acc_b)          # chunk invariance, bit for bit
    assert torch.isfinite(rgb_a).all() and torch.isfinite(acc_a).all()
    assert float(acc_a.min()) >= 0.0 and float(acc_a.max()) <= 1.0 + 1e-5
    assert float(rgb_a.min()) >= 0.0 and float(rgb_a.max()) <= 1.0 + 1e-5
    # jitter and noise are functions of the seed only
    sub = rb[::37].contiguous()
    kw2 = dict(num_coarse=64, num_fine=128, perturb=True, radiance_field_noise_std=0.2)
    with torch.no_grad():
        r1 = nk.render_rays(sub, coarse, fine, seed=5, **kw2)[0]
        r2 = nk.render_rays(sub, coarse, fine, seed=5, **kw2)[0]
        r3 = nk.render_rays(sub, coarse, fine, seed=6, **kw2)[0]
    assert torch.equal(r1, r2) and not torch.equal(r1, r3)
    return rb, rgb_a


def _subset_vs_oracle(nk, scene, k, rb, coarse, fine, oarch, wc, wf, dev, n=768, mean_tol=3e-3):
    g = torch.Generator().manual_seed(3)
    sel = torch.randperm(scene.H * scene.W, generator=g)[:n]
    ref_rb = _oracle_rays(scene, k, sel)
    got_rb = rb[sel.to(dev)].cpu()
    np.testing.assert_allclose(got_rb.numpy(), ref_rb.numpy(), rtol=3e-5, atol=3e-6)
    co, fo = O.NeRFNet.from_flat(oarch, wc), O.NeRFNet.from_flat(oarch, wf)
    with torch.no_grad():
        ref = O.render_rays(ref_rb, co, fo, O.RenderOpts(64, 128))
        got = nk.render_rays(rb[sel.to(dev)].contiguous(), coarse, fine, num_coarse=64, num_fine=128)
    d = (got[0].cpu() - ref["rgb_fine"]).abs()
    print(f"{n} rays vs fp32 oracle: mean |d rgb| {float(d.mean()):.5f}, max {float(d.max()):.4f}")
    assert float(d.mean()) < mean_tol and float(d.max()) < 0.15
    assert float((got[2].cpu() - ref["acc_fine"]).abs().mean()) < mean_tol


def test_forward_grid_1280x720(nk, golden_dir):
    from nerf_kinematics_amd import datasets
    dev = torch.device("cuda:0")
    scene = datasets.forward_grid_scene(None, 3, 1280, 720, dev, holdout=8)
    assert scene.images.shape == (9, 720, 1280, 3) and scene.ndc and (scene.near, scene.far) == (0.0, 1.0)
    assert list(scene.i_val) == [0, 8] and len(scene.i_train) == 7
    # the photographs show the object: not empty, not saturated
    m = scene.images[4].float().mean() / 255.0
    assert 0.02 < float(m) < 0.6
    # the lego networks (ten position frequencies) rendered through the NDC rays of the centre view
    w = np.load(os.path.join(golden_dir, "lego_lowres3_weights.npz"))
    wc, wf = torch.from_numpy(w["coarse"]), torch.from_numpy(w["fine"])
    coarse, fine = nk.NeRFModel(nk.ARCH_LEGO_CKPT, wc.to(dev)), nk.NeRFModel(nk.ARCH_LEGO_CKPT, wf.to(dev))
    rb, _ = _full_frame_properties(nk, scene, 4, coarse, fine, dev)
    _subset_vs_oracle(nk, scene, 4, rb, coarse, fine, O.ARCH_LEGO_CKPT, wc, wf, dev)
    # the fused sampler on this scene: pixels, rays and colours of one training iteration
    smp = datasets.RaySampler(scene, 4096, seed=7)
    rays, target = smp.next(0)
    k = smp.view_of(0)
    pix = smp.pixels.long()
    assert torch.equal(rays, rb[pix]) if k == 4 else True
    assert torch.allclose(target, scene.images[k].reshape(-1, 3)[pix].float() / 255.0, rtol=0, atol=1e-7)


def test_forward_grid_trains(nk):
    """A short NDC training run on the grid scene at 1280 x 720: held-out PSNR rises.  Density
    noise 1.0 as forward-facing experiments of this lineage use: without it the first iterations are
    bistable (about one trajectory in five settles on a constant image, 13.8 dB, and which one is
    decided by the last bit of a gradient sum)."""
    from nerf_kinematics_amd import datasets
    from nerf_kinematics_amd.metrics import mse2psnr
    from nerf_kinematics_amd.trainer import Trainer, TrainOpts
    dev = torch.device("cuda:0")
    scene = datasets.forward_grid_scene(None, 3, 1280, 720, dev, holdout=8)
    opts = TrainOpts(num_coarse=64, num_fine=64, num_random_rays=4096, lr=5e-4, radiance_field_noise_std=1.0, seed=1)
    tr = Trainer(nk.ARCH_RUN_NERF, nk.ARCH_RUN_NERF, opts, dev)
    smp = datasets.RaySampler(scene, 4096, seed=1)

    def psnr():
        k = int(scene.i_val[0])
        sel = slice(0, None, 9)
        out = tr.render(scene.rays(k, dev)[sel].contiguous())
        return mse2psnr(float(((out["rgb_fine"] - scene.target(k)[sel]) ** 2).mean()))
    p0 = psnr()
    for i in range(400):
        tr.step(*smp.next(i))
    p1 = psnr()
    print(f"forward grid 1280x720: held-out PSNR {p0:.2f} -> {p1:.2f} dB after 400 iterations")
    assert p1 > p0 + 5.0 and p1 > 17.0


def test_wheel_capture_1280x720(nk, golden_dir, tmp_path):
    """The robot capture at full resolution (downsample_factor 1) with the reference's trained wheel
    weights: loader, NDC rays and the six-frequency network at 1280 x 720."""
    import shutil
    from nerf_kinematics_amd import datasets
    dev = torch.device("cuda:0")
    cap = tmp_path / "capture"
    shutil.copytree(os.path.join(golden_dir, "wheel_real_robot"), cap)
    r = subprocess.run([sys.executable, "-m", "nerf_kinematics_amd.poses", "--poses", str(cap / "poses.txt"),
                        "--image_folder", str(cap / "images_robot"), "--recenter", "--output", str(cap / "transforms.json")],
                       env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    cfg = {"dataset": {"type": "robot", "basedir": str(cap), "downsample_factor": 1, "llffhold": 8, "near": 0, "far": 1,
                       "no_ndc": False}, "nerf": {"train": {"white_background": False}}}
    scene = datasets.load_scene(cfg, dev)
    assert (scene.W, scene.H) == (1280, 720) and scene.ndc and scene.images.shape[0] == 12
    assert list(scene.i_val) == [0, 8]
    # the 8x down-sampled scene keeps every 8th row and column of these images
    cfg["dataset"]["downsample_factor"] = 8
    small = datasets.load_scene(cfg, dev)
    assert (small.W, small.H) == (160, 90) and torch.equal(small.images, scene.images[:, ::8, ::8])
    assert abs(small.fx - scene.fx / 8) < 1e-9
    w = np.load(os.path.join(golden_dir, "wheel_robot6_weights.npz"))
    wc, wf = torch.from_numpy(w["coarse"]), torch.from_numpy(w["fine"])
    coarse, fine = nk.NeRFModel(nk.ARCH_WHEEL_CKPT, wc.to(dev)), nk.NeRFModel(nk.ARCH_WHEEL_CKPT, wf.to(dev))
    k = int(scene.i_val[0])
    rb, _ = _full_frame_properties(nk, scene, k, coarse, fine, dev)
    # (the wheel field is an over-fitted, near-saturated function of position: bf16 operand rounding
    # moves its renders about four times as much as the lego field's, hence the wider mean bound)
    _subset_vs_oracle(nk, scene, k, rb, coarse, fine, O.ARCH_WHEEL_CKPT, wc, wf, dev, mean_tol=8e-3)


def test_lego_800x800_single_gpu(nk, golden_dir):
    """The full-resolution lego configuration (800 x 800; meant to be sharded over 8 GPUs) on ONE
    GPU: the whole frame rendered with the reference's trained lego networks -- chunk invariance,
    bounds, seed determinism over all 640 000 rays, the oracle on a ray subset -- and the row-block
    sharding arithmetic that splits such a frame over 8 ranks."""
    import math
    from nerf_kinematics_amd import datasets, dist as nd
    dev = torch.device("cuda:0")
    focal = 0.5 * 800 / math.tan(0.5 * datasets.LEGO_CAMERA_ANGLE_X)
    scene = datasets.Scene(None, datasets.spherical_pose(45.0, -30.0, 4.0)[None], 800, 800, focal, focal, 400.0, 400.0, 2.0, 6.0)
    w = np.load(os.path.join(golden_dir, "lego_lowres3_weights.npz"))
    wc, wf = torch.from_numpy(w["coarse"]), torch.from_numpy(w["fine"])
    coarse, fine = nk.NeRFModel(nk.ARCH_LEGO_CKPT, wc.to(dev)), nk.NeRFModel(nk.ARCH_LEGO_CKPT, wf.to(dev))
    rb = scene.rays(0, dev)
    assert rb.shape == (640000, 11)
    kw = dict(num_coarse=64, num_fine=128)
    with torch.no_grad():
        full = torch.cat([nk.render_rays(rb[s:s + 131072], coarse, fine, **kw)[0] for s in range(0, 640000, 131072)])
        # the eight row blocks an 8-rank job would render, each on its own, concatenated
        blocks = [nd.row_block(640000, r, 8) for r in range(8)]
        assert blocks[0] == slice(0, 80000) and blocks[7] == slice(560000, 640000)
        sharded = torch.cat([nk.render_rays(rb[b].contiguous(), coarse, fine, **kw)[0] for b in blocks])
    assert torch.equal(full, sharded)
    assert torch.isfinite(full).all() and float(full.min()) >= 0.0 and float(full.max()) <= 1.0 + 1e-5
    img = full.reshape(800, 800, 3)
    assert float(img[350:450, 350:450].mean()) > 0.05 and float(img[:40, :40].mean()) < 0.02     # object in the middle, (nearly) empty corner
    _subset_vs_oracle(nk, scene, 0, rb, coarse, fine, O.ARCH_LEGO_CKPT, wc, wf, dev, n=512)
