"""GPU parity tests of networks WITHOUT view directions (``use_viewdirs: False``,
/root/reference/trained_models/nerf-pytorch/lego-lowres3/config.yml:30,42,55 are the keys): either lineage
with the trunk followed by one four-row linear layer (rgb_raw, sigma_raw), ray rows of 8 columns.

Parity unpinned: the reference ships no such network; the oracle follows the public nerf-pytorch
model classes (``fc_out`` / ``output_linear``).  Tolerances as for the networks with a direction
branch: raw outputs 1e-2 absolute against the bf16-emulating oracle, per-tensor relative L2 of the
gradient 3e-2 (MLP alone) / 5e-2 (whole render_rays).
"""
import math
import os
import subprocess
import sys

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


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def archs(nk):
    A = nk.Arch
    return {
        "flexible": A(4, 128, -1, 64, 10, 4, False, True, "flexible", use_viewdirs=False),
        "flexible_L6": A(4, 128, -1, 64, 6, 4, False, True, "flexible", use_viewdirs=False),
        "run_nerf": A(8, 256, 4, 128, 10, 4, True, False, "run_nerf", use_viewdirs=False),
        "run_nerf_with_dirs": nk.ARCH_RUN_NERF,
    }


def oarch(a):
    return O.Arch(a.depth, a.width, a.skip, a.dir_width, a.n_freq_xyz, a.n_freq_dir, a.relu_first, a.relu_feat, a.lineage,
                  a.use_viewdirs)


def rays(R, seed, use_viewdirs=False):
    g = torch.Generator().manual_seed(seed)
    focal = 0.5 * 16 / math.tan(0.5 * 0.6911112070083618)
    ro, rd = O.get_rays(16, 16, focal, O.spherical_pose(45.0, -30.0, 4.0))
    rb = O.make_ray_batch(ro, rd, 2.0, 6.0, use_viewdirs=use_viewdirs)
    return rb[torch.randperm(256, generator=g)[:R]].contiguous(), g


def per_tensor_rel_err(arch, got, ref):
    out, off = {}, 0
    for name, (o, i) in arch.param_shapes():
        for suffix, cnt in ((".weight", o * i), (".bias", o)):
            a, b = got[off:off + cnt], ref[off:off + cnt]
            out[name + suffix] = float((a - b).norm() / (b.norm() + 1e-12)), float(b.norm())
            off += cnt
    assert off == got.numel() == ref.numel()
    return out


def check_grads(arch, got, ref, tol):
    errs = per_tensor_rel_err(arch, got.cpu(), ref)
    total = float(ref.norm())
    bad = {k: v for k, v in errs.items() if v[0] > tol and v[1] > 1e-6 * total}
    assert not bad, bad
    assert float((got.cpu() - ref).norm() / ref.norm()) < tol / 2
    # every tensor of the network receives a gradient (the output layer's four rows included)
    assert all(v[1] > 0 for v in errs.values())


def test_parameter_layout(nk):
    from nerf_kinematics_amd import lib as nklib
    L = nk.load()
    a = archs(nk)
    for key, last, n in (("flexible", "fc_out", 63 * 128 + 128 + 3 * (128 * 128 + 128) + 4 * 128 + 4),
                         ("run_nerf", "output_linear", None)):
        arch = a[key]
        names = [nm for nm, _ in arch.param_shapes()]
        assert names[-1] == last and dict(arch.param_shapes())[last] == (4, arch.width)
        assert not any("dir" in nm or "views" in nm or "alpha" in nm or "feat" in nm or "rgb" in nm for nm in names)
        assert nklib.check(L.nk_num_params(arch.to_c())) == arch.num_params() == oarch(arch).num_params()
        if n is not None:
            assert arch.num_params() == n
        # the bf16 fragment image: the forward streams keep the stage sequence of a network with a direction
        # branch, the transposed stream loses the two stages of that branch (WD/32 + W/32 * WD/16 blocks)
        with_dirs = nk.Arch(arch.depth, arch.width, arch.skip, arch.dir_width, arch.n_freq_xyz, arch.n_freq_dir,
                            arch.relu_first, arch.relu_feat, arch.lineage)
        W, WD = arch.width, arch.dir_width
        assert nklib.check(L.nk_packed_bytes(with_dirs.to_c())) - nklib.check(L.nk_packed_bytes(arch.to_c())) == \
            (WD // 32 + (W // 32) * (WD // 16)) * 1024


@pytest.mark.parametrize("which,R,n", [("flexible", 20, 64), ("flexible_L6", 7, 33), ("run_nerf", 12, 192)])
def test_mlp_forward_and_backward_match_the_oracle(nk, dev, which, R, n):
    """Both forward kernels (the saving one on 32x32x16, the inference one on 16x16x32) and the
    backward chain + weight-gradient products on a network that ends in the four-row output layer."""
    arch = archs(nk)[which]
    flat = O.NeRFNet.init_flat(oarch(arch), 7)
    rb, g = rays(R, 11)
    assert rb.shape[1] == 8
    z = torch.sort(torch.rand(R, n, generator=g) * 4 + 2, -1)[0]
    d_raw = torch.randn(R, n, 4, generator=g)
    model = nk.NeRFModel(arch, flat.to(dev))
    act = nk.alloc_activations(model, R, n)
    raw = nk.run_network(model, rb.to(dev), z.to(dev), act_save=act)
    raw_inf = nk.run_network(model, rb.to(dev), z.to(dev))
    grad = nk.mlp_backward(model, act, d_raw.to(dev))
    fl = flat.clone().requires_grad_(True)
    net = O.NeRFNet.from_flat(oarch(arch), fl, emulate_bf16=True)
    ref_raw = O._run_net(net, rb, z)
    (ref_raw * d_raw).sum().backward()
    assert float((raw.cpu() - ref_raw.detach()).abs().max()) < 1e-2
    assert float((raw_inf.cpu() - ref_raw.detach()).abs().max()) < 1e-2
    assert float((raw_inf - raw).abs().max()) < 2e-3        # same bf16 operands, another accumulation order
    # and against the fp32 network: bf16 operand rounding only
    ref32 = O._run_net(O.NeRFNet.from_flat(oarch(arch), flat), rb, z)
    assert float((raw.cpu() - ref32).abs().max()) < 5e-2
    check_grads(arch, grad, fl.grad, 3e-2)


@pytest.mark.parametrize("which", ["flexible", "run_nerf", "mixed"])
def test_render_rays_gradients(nk, dev, which):
    """Whole render_rays (coarse + fine, perturb + noise) on 8-column rays: maps and gradient of the
    NeRF loss against the oracle.  "mixed": a coarse network without view directions under a fine
    network with them (11-column rows, which the coarse network reads 8 columns of)."""
    a = archs(nk)
    arch_c = a["run_nerf"] if which == "run_nerf" else a["flexible"]
    arch_f = a["run_nerf_with_dirs"] if which == "mixed" else arch_c
    R = 40 if which == "flexible" else 24
    fc, ff = O.NeRFNet.init_flat(oarch(arch_c), 1), O.NeRFNet.init_flat(oarch(arch_f), 2)
    rb, g = rays(R, 21, use_viewdirs=which == "mixed")
    target = torch.rand(R, 3, generator=g)
    kw = dict(perturb=True, radiance_field_noise_std=0.2, seed=5)
    c = nk.NeRFModel(arch_c, fc.to(dev)); f = nk.NeRFModel(arch_f, ff.to(dev))
    c.flat.requires_grad_(True); f.flat.requires_grad_(True)
    out = nk.render_rays(rb.to(dev), c, f, num_coarse=64, num_fine=128, **kw)
    tg = target.to(dev)
    loss = ((out[3] - tg) ** 2).mean() + ((out[0] - tg) ** 2).mean()
    loss.backward()
    fco, ffo = fc.clone().requires_grad_(True), ff.clone().requires_grad_(True)
    co = O.NeRFNet.from_flat(oarch(arch_c), fco, emulate_bf16=True)
    fo = O.NeRFNet.from_flat(oarch(arch_f), ffo, emulate_bf16=True)
    ref = O.render_rays(rb, co, fo, O.RenderOpts(64, 128, **kw))
    lref = O.nerf_loss(ref, target)
    lref.backward()
    assert float((out[0].detach().cpu() - ref["rgb_fine"].detach()).abs().max()) < 5e-3
    assert float((out[3].detach().cpu() - ref["rgb_coarse"].detach()).abs().max()) < 5e-3
    assert abs(float(loss.detach()) - float(lref.detach())) < 2e-3 * float(lref.detach()) + 1e-5
    check_grads(arch_c, c.flat.grad, fco.grad, 5e-2)
    check_grads(arch_f, f.flat.grad, ffo.grad, 5e-2)


def test_large_batch_gradients(nk, dev):
    """8x256 pair without view directions after 200 training iterations, one ragged batch of 5 507 rays
    (1 057 344 fine points): above the size where the weight-gradient products are launched network by
    network in about 256 K-slices each, so the output layer's own product (1 x 8 tiles, dOut against
    h_{D-1}) runs with long slices and a short last one.  Oracle evaluated in pieces of 1 024 rays."""
    from nerf_kinematics_amd.datasets import RaySampler, teacher_scene
    from nerf_kinematics_amd.trainer import Trainer, TrainOpts
    from nerf_kinematics_amd import lib as nklib
    arch = archs(nk)["run_nerf"]
    R = 5507
    assert R * 192 > nklib.PAIR_GEMM_MAX_POINTS
    scene = teacher_scene(None, 8, 0, 100, dev)
    tr = Trainer(arch, arch, TrainOpts(num_coarse=64, num_fine=128, num_random_rays=1024, perturb=True,
                                       radiance_field_noise_std=0.2, lr=5e-4, seed=42), dev)
    smp = RaySampler(scene, 1024, seed=3)
    l0 = float(tr.step(*smp.next(0)))
    for i in range(1, 200):
        l1 = float(tr.step(*smp.next(i)))
    assert l1 < 0.7 * l0                        # the pair trains
    rb, target = RaySampler(scene, R, seed=8).next(200)
    rb8 = rb[:, :8].contiguous()               # the rows the reference builds for such networks
    fc, ff = tr.coarse.flat.detach().clone(), tr.fine.flat.detach().clone()
    kw = dict(perturb=True, radiance_field_noise_std=0.2, seed=77)
    c, f = nk.NeRFModel(arch, fc.clone()), nk.NeRFModel(arch, ff.clone())
    c.flat.requires_grad_(True); f.flat.requires_grad_(True)
    out = nk.render_rays(rb8, c, f, num_coarse=64, num_fine=128, **kw)
    loss = ((out[3] - target) ** 2).mean() + ((out[0] - target) ** 2).mean()
    loss.backward()
    fco, ffo = fc.cpu().requires_grad_(True), ff.cpu().requires_grad_(True)
    rbc, tc, total = rb8.cpu(), target.cpu(), 0.0
    for s in range(0, R, 1024):
        co = O.NeRFNet.from_flat(oarch(arch), fco, emulate_bf16=True)
        fo = O.NeRFNet.from_flat(oarch(arch), ffo, emulate_bf16=True)
        o = O.render_rays(rbc[s:s + 1024], co, fo, O.RenderOpts(64, 128, **kw), ray_offset=s)
        t = tc[s:s + 1024]
        part = (((o["rgb_coarse"] - t) ** 2).sum() + ((o["rgb_fine"] - t) ** 2).sum()) / (3 * R)
        part.backward()
        total += float(part.detach())
    print(f"loss: HIP {float(loss.detach()):.6f}, oracle {total:.6f}; whole-gradient relative L2 coarse "
          f"{float((c.flat.grad.cpu() - fco.grad).norm() / fco.grad.norm()):.4f}, fine "
          f"{float((f.flat.grad.cpu() - ffo.grad).norm() / ffo.grad.norm()):.4f}")
    assert abs(float(loss.detach()) - total) < 2e-3 * total + 1e-5
    check_grads(arch, c.flat.grad, fco.grad, 5e-2)
    check_grads(arch, f.flat.grad, ffo.grad, 5e-2)


def test_columns_8_to_10_are_never_read(nk, dev):
    """8-column rows and 11-column rows with NaN in the direction columns give the same bits, in
    inference and in training; a network WITH a direction branch refuses 8-column rows."""
    arch = archs(nk)["flexible"]
    c = nk.NeRFModel(arch, O.NeRFNet.init_flat(oarch(arch), 3).to(dev))
    f = nk.NeRFModel(arch, O.NeRFNet.init_flat(oarch(arch), 4).to(dev))
    rb8, _ = rays(100, 2)
    rb11 = torch.cat([rb8, torch.full((100, 3), float("nan"))], 1)
    same = lambda x, y: torch.equal(torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(y, nan=-7.0))
    with torch.no_grad():
        a = nk.render_rays(rb8.to(dev), c, f, num_coarse=64, num_fine=64)
        b = nk.render_rays(rb11.to(dev), c, f, num_coarse=64, num_fine=64)
    assert all(same(x, y) for x, y in zip(a, b)) and torch.isfinite(a[0]).all()
    grads = []
    for rb in (rb8, rb11):
        for m in (c, f):
            m.flat.requires_grad_(True); m.flat.grad = None
        out = nk.render_rays(rb.to(dev), c, f, num_coarse=64, num_fine=64, perturb=True, seed=1)
        (out[0].sum() + out[3].sum()).backward()
        grads.append((c.flat.grad.clone(), f.flat.grad.clone()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
    assert torch.isfinite(grads[0][0]).all() and float(grads[0][1].abs().sum()) > 0
    with_dirs = nk.NeRFModel(nk.ARCH_LEGO_CKPT, O.NeRFNet.init_flat(O.ARCH_LEGO_CKPT, 3).to(dev))
    with pytest.raises(ValueError, match="8-column"):
        nk.render_rays(rb8.to(dev), with_dirs, None, num_coarse=64, num_fine=0)


@pytest.mark.parametrize("fmt", ["fp8", "fp8bf8"])
def test_8bit_saved_tensors(nk, dev, fmt):
    """The experimental saved-tensor formats on a network without view directions: same maps, and a
    gradient that matches the oracle's emulation of the format."""
    from nerf_kinematics_amd.render import dz_scale
    arch = archs(nk)["run_nerf"]
    R = 24
    fc, ff = O.NeRFNet.init_flat(oarch(arch), 1), O.NeRFNet.init_flat(oarch(arch), 2)
    rb, g = rays(R, 9)
    target = torch.rand(R, 3, generator=g)
    kw = dict(perturb=True, radiance_field_noise_std=0.2, seed=5)
    res = {}
    for f_ in ("bf16", fmt):
        c = nk.NeRFModel(arch, fc.to(dev)); f = nk.NeRFModel(arch, ff.to(dev))
        c.flat.requires_grad_(True); f.flat.requires_grad_(True)
        out = nk.render_rays(rb.to(dev), c, f, num_coarse=64, num_fine=128, act_format=f_, **kw)
        (((out[3] - target.to(dev)) ** 2).mean() + ((out[0] - target.to(dev)) ** 2).mean()).backward()
        res[f_] = (out[0].detach().clone(), c.flat.grad.clone(), f.flat.grad.clone())
    assert torch.equal(res["bf16"][0], res[fmt][0])
    fco, ffo = fc.clone().requires_grad_(True), ff.clone().requires_grad_(True)
    S = dz_scale(R)
    co = O.NeRFNet.from_flat(oarch(arch), fco, emulate_bf16=True, act_format=fmt, dz_scale=S)
    fo = O.NeRFNet.from_flat(oarch(arch), ffo, emulate_bf16=True, act_format=fmt, dz_scale=S)
    O.nerf_loss(O.render_rays(rb, co, fo, O.RenderOpts(64, 128, **kw)), target).backward()
    tol = 5e-2 if fmt == "fp8" else 1e-1
    check_grads(arch, res[fmt][1], fco.grad, tol)
    check_grads(arch, res[fmt][2], ffo.grad, tol)


def test_training_and_checkpoint_through_the_cli(nk, tmp_path):
    """`run_nerf.py` on a config with use_viewdirs: False: trains, the loss falls, the checkpoint holds
    the reference's key names for such a network and loads back for a render."""
    import json
    import yaml
    with open(os.path.join(ROOT, "tests", "golden", "lego-lowres3_config.yml")) as fh:
        cfg = yaml.safe_load(fh)
    cfg["dataset"] = {"type": "teacher", "near": 2, "far": 6, "train_views": 6, "val_views": 2, "resolution": 48}
    cfg["experiment"].update(id="novd", logdir=str(tmp_path), train_iters=60, print_every=10, validate_every=30, save_every=30)
    cfg["nerf"]["use_viewdirs"] = False
    cfg["models"]["coarse"]["use_viewdirs"] = False
    cfg["models"]["fine"]["use_viewdirs"] = False
    p = tmp_path / "exp.yml"
    p.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, PYTHONPATH=ROOT)
    run = lambda *extra: subprocess.run([sys.executable, os.path.join(ROOT, "run_nerf.py"), "--config", str(p), *extra],
                                        env=env, capture_output=True, text=True, timeout=300)
    r = run()
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = tmp_path / "novd"
    recs = [json.loads(l) for l in (out / "log.jsonl").read_text().splitlines()]
    train = [r_["train"] for r_ in recs if "train" in r_]
    assert train[-1]["loss"] < 0.7 * train[0]["loss"]
    ck = torch.load(out / "checkpoint00059.ckpt", map_location="cpu", weights_only=True)
    keys = list(ck["model_fine_state_dict"].keys())
    assert keys == ["layer1.weight", "layer1.bias", "layers_xyz.0.weight", "layers_xyz.0.bias", "layers_xyz.1.weight",
                    "layers_xyz.1.bias", "layers_xyz.2.weight", "layers_xyz.2.bias", "fc_out.weight", "fc_out.bias"]
    assert tuple(ck["model_fine_state_dict"]["fc_out.weight"].shape) == (4, 128)
    r = run("--load-checkpoint", str(out / "checkpoint00059.ckpt"), "--render-only")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert len(list((out / "render").glob("*.png"))) == 2
    # a file whose two switches disagree is refused before anything runs
    cfg["models"]["fine"]["use_viewdirs"] = True
    p.write_text(yaml.safe_dump(cfg))
    r = run()
    assert r.returncode != 0 and "use_viewdirs" in (r.stdout + r.stderr)
