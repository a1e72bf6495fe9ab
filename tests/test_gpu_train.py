"""GPU tests of the callers around render_rays: the fused training-ray sampler against the
oracle, checkpoint resume, a short optimisation run on a synthetic scene (held-out PSNR must
rise), and the command-line entry point end to end."""
import json
import math
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


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def teacher(nk, dev):
    from nerf_kinematics_amd import datasets
    return datasets.teacher_scene(None, 12, 3, 64, dev, num_coarse=64, num_fine=64)


@pytest.mark.parametrize("channels,white", [(3, False), (4, False), (4, True)])
def test_training_ray_sampler_matches_oracle(nk, dev, channels, white):
    """Pixel draws bit-exact, colours bit-exact, rays rtol 1e-5 (same bar as get_rays)."""
    from nerf_kinematics_amd import datasets
    H, W, R = 37, 53, 1000
    rng = np.random.default_rng(channels)
    imgs = rng.integers(0, 256, (3, H, W, channels), dtype=np.uint8)
    poses = np.stack([datasets.spherical_pose(40.0 * k, -30.0, 4.0) for k in range(3)])
    focal = 0.5 * W / math.tan(0.5 * 0.6911112070083618)
    sc = datasets.Scene(torch.from_numpy(imgs).to(dev), poses, H, W, focal, focal, W * 0.5, H * 0.5, 2.0, 6.0, False,
                        white, np.arange(3))
    smp = datasets.RaySampler(sc, R, seed=1234, rank=1, world=2)
    seen = set()
    for it in (0, 1, 5, 2 ** 20):
        rb, tgt = smp.next(it)
        k = smp.view_of(it)
        seen.add(k)
        assert smp.counter_of(it) == (it * 2 + 1) * R
        pix, rb_ref, tgt_ref = O.sample_training_rays(H, W, focal, torch.from_numpy(poses[k]), imgs[k], white, 2.0, 6.0,
                                                      1234, smp.counter_of(it), R)
        assert np.array_equal(smp.pixels.cpu().numpy().astype(np.int64), pix)
        assert pix.min() >= 0 and pix.max() < H * W
        assert torch.equal(tgt.cpu(), tgt_ref)
        np.testing.assert_allclose(rb.cpu().numpy(), rb_ref.numpy(), rtol=1e-5, atol=1e-6)
    assert seen <= {0, 1, 2}
    # draws cover the image roughly uniformly (size-independent property of the generator)
    big = datasets.RaySampler(sc, 200000, seed=7)
    big.next(3)
    hist = torch.bincount(big.pixels.long(), minlength=H * W).float()
    assert hist.min() > 0.5 * 200000 / (H * W) and hist.max() < 1.6 * 200000 / (H * W)


def test_sampler_rejects_bad_input(nk, dev):
    from nerf_kinematics_amd import datasets
    sc = datasets.Scene(torch.zeros(1, 4, 4, 2, dtype=torch.uint8, device=dev), np.eye(4, dtype=np.float32)[None], 4, 4,
                        5.0, 5.0, 2.0, 2.0, 2.0, 6.0, i_train=np.arange(1))
    with pytest.raises(ValueError, match="channels"):
        datasets.RaySampler(sc, 16).next(0)
    sc.images = sc.images.cpu()
    with pytest.raises(ValueError, match="GPU"):
        datasets.RaySampler(sc, 16)


def test_teacher_scene_is_the_lego_probe(nk, dev, teacher):
    """The synthetic photographs show the object: opaque centre, dark corners, held-out split."""
    assert tuple(teacher.images.shape) == (15, 64, 64, 3)
    img = teacher.images.float() / 255
    assert float(img[:, 26:38, 26:38].mean()) > 0.15 and float(img[:, :6, :6].mean()) < 0.02
    assert teacher.i_train.tolist() == list(range(12)) and teacher.i_val.tolist() == [12, 13, 14]


def _trainer(nk, dev, rays=1024, seed=42, fine=64):
    from nerf_kinematics_amd.trainer import Trainer, TrainOpts
    opts = TrainOpts(num_coarse=64, num_fine=fine, num_random_rays=rays, perturb=True, radiance_field_noise_std=0.2,
                     lr=5e-3, seed=seed)
    return Trainer(nk.ARCH_LEGO_CKPT, nk.ARCH_LEGO_CKPT, opts, dev)


def _heldout_psnr(tr, scene):
    from nerf_kinematics_amd.metrics import mse2psnr
    vals = []
    for k in scene.i_val:
        out = tr.render(scene.rays(int(k), tr.device), opts=dict(num_coarse=64, num_fine=64))
        vals.append(mse2psnr(float(((out["rgb_fine"] - scene.target(int(k))) ** 2).mean())))
    return float(np.mean(vals))


def test_short_training_run_raises_heldout_psnr(nk, dev, teacher):
    """400 iterations of the 4x128 pair on 12 synthetic views: the loss falls and held-out PSNR
    rises by several dB (the end-to-end check that gradients, Adam and the re-pack cooperate)."""
    from nerf_kinematics_amd.datasets import RaySampler
    tr = _trainer(nk, dev)
    smp = RaySampler(teacher, 1024, seed=42)
    p0 = _heldout_psnr(tr, teacher)
    losses = []
    for i in range(400):
        rb, tgt = smp.next(i)
        loss = tr.step(rb, tgt)
        if i % 50 == 0 or i == 399:
            losses.append(float(loss))
    p1 = _heldout_psnr(tr, teacher)
    print(f"held-out PSNR {p0:.2f} -> {p1:.2f} dB, losses {losses}")
    assert all(math.isfinite(v) for v in losses)
    assert losses[-1] < 0.5 * losses[0]
    assert p1 > p0 + 4.0 and p1 > 17.0


def test_training_psnr_tracks_the_fp32_oracle(nk, dev):
    """The same optimisation run through the HIP path (bf16 MFMA operands) and through the fp32 CPU
    restatement: identical initial parameters, identical rays and colours every iteration (the GPU
    sampler's), identical jitter / noise seeds, torch Adam with the same schedule.  The first loss
    agrees to 1e-4; after that the two runs are two trajectories of a chaotic optimisation, so
    held-out PSNR is compared as a mean over nine checkpoints (iterations 300..700): measured
    differences are -0.02 dB (seed 42) and +0.12 dB (seed 7); over three longer runs (105 checkpoints)
    +0.03 dB with a standard error of 0.02 dB, single checkpoints up to 0.57 dB apart in either
    direction; bounds 0.3 dB on the mean, 1 dB on any checkpoint."""
    from nerf_kinematics_amd import datasets
    from nerf_kinematics_amd.metrics import mse2psnr
    from nerf_kinematics_amd.trainer import Trainer, TrainOpts, _init_flat, lr_at
    Nc, Nf, R, seed = 32, 32, 512, 42
    scene = datasets.teacher_scene(None, 12, 3, 32, dev, num_coarse=64, num_fine=64)
    arch = nk.ARCH_LEGO_CKPT
    oa = O.Arch(arch.depth, arch.width, arch.skip, arch.dir_width, arch.n_freq_xyz, arch.n_freq_dir, arch.relu_first,
                arch.relu_feat, arch.lineage)
    opts = TrainOpts(num_coarse=Nc, num_fine=Nf, num_random_rays=R, perturb=True, radiance_field_noise_std=0.2,
                     lr=1e-3, seed=seed)
    tr = Trainer(arch, arch, opts, dev)
    smp = datasets.RaySampler(scene, R, seed=seed)
    fc = _init_flat(arch, seed).clone().requires_grad_(True)
    ff = _init_flat(arch, seed + 1).clone().requires_grad_(True)
    opt = torch.optim.Adam([fc, ff], lr=opts.lr, betas=opts.betas, eps=opts.eps)
    val = [(scene.rays(int(k), dev), scene.target(int(k))) for k in scene.i_val]

    def psnr_hip():
        return float(np.mean([mse2psnr(float(((tr.render(rb, opts=dict(num_coarse=Nc, num_fine=Nf))["rgb_fine"] - t) ** 2).mean()))
                              for rb, t in val]))

    @torch.no_grad()
    def psnr_oracle():
        co, fo = O.NeRFNet.from_flat(oa, fc), O.NeRFNet.from_flat(oa, ff)
        ro = O.RenderOpts(Nc, Nf, perturb=False, radiance_field_noise_std=0.0)
        return float(np.mean([mse2psnr(float(((O.render_rays(rb.cpu(), co, fo, ro)["rgb_fine"] - t.cpu()) ** 2).mean()))
                              for rb, t in val]))

    hip, ora = [], []
    for i in range(700):
        rb, tgt = smp.next(i)
        lh = float(tr.step(rb, tgt))
        co, fo = O.NeRFNet.from_flat(oa, fc), O.NeRFNet.from_flat(oa, ff)
        ro = O.RenderOpts(Nc, Nf, perturb=True, radiance_field_noise_std=0.2, seed=(seed + 0x9E3779B1 * (i + 1)) & 0xFFFFFFFF)
        lo = O.nerf_loss(O.render_rays(rb.cpu(), co, fo, ro), tgt.cpu())
        if i == 0:
            assert abs(lh - float(lo.detach())) < 1e-4 * lh
        for g in opt.param_groups:
            g["lr"] = lr_at(i, opts)
        opt.zero_grad(); lo.backward(); opt.step()
        if i + 1 >= 300 and (i + 1) % 50 == 0:
            hip.append(psnr_hip()); ora.append(psnr_oracle())
    d = np.array(hip) - np.array(ora)
    print(f"held-out PSNR, mean over {len(d)} checkpoints: HIP {np.mean(hip):.3f} dB, oracle {np.mean(ora):.3f} dB, "
          f"difference {d.mean():+.3f} dB (single checkpoints {d.min():+.3f} .. {d.max():+.3f})")
    assert np.mean(ora) > 17.0
    assert abs(d.mean()) < 0.3 and np.abs(d).max() < 1.0


def test_one_call_step_equals_the_stage_by_stage_step(nk, dev, teacher):
    """Trainer.step through nk_render_rays_fwd/_bwd (which, on a batch this small, runs the
    weight-gradient products of both networks in one shared launch) against the same step issued
    stage by stage through the per-stage entry points (one nk_mlp_backward per network): same
    loss, same gradient up to the summation order of the K-slices."""
    from nerf_kinematics_amd import lib as nklib
    from nerf_kinematics_amd.datasets import RaySampler
    rb, tgt = RaySampler(teacher, 1024, seed=3).next(0)
    a, b = _trainer(nk, dev, seed=11), _trainer(nk, dev, seed=11)
    la = float(a.step(rb, tgt))
    o = a.ropts                                          # holds the seed the step used
    opts = nk.make_opts(o.num_coarse, o.num_fine, bool(o.perturb), bool(o.lindisp), o.radiance_field_noise_std,
                        bool(o.white_background), o.seed)
    R, Nc, N = rb.shape[0], o.num_coarse, o.num_coarse + o.num_fine
    z_c = nk.stratified_z(rb, opts)
    act_c = nk.alloc_activations(b.coarse, R, Nc)
    raw_c = nk.run_network(b.coarse, rb, z_c, act_c)
    out_c = nk.raw2outputs(raw_c, z_c, rb, opts, nklib.STREAM_NOISE_COARSE)
    z_all, _ = nk.sample_pdf(z_c, out_c["weights"], opts)
    act_f = nk.alloc_activations(b.fine, R, N)
    raw_f = nk.run_network(b.fine, rb, z_all, act_f)
    out_f = nk.raw2outputs(raw_f, z_all, rb, opts, nklib.STREAM_NOISE_FINE)
    loss, dc, df = nk.mse_loss(out_c["rgb"], out_f["rgb"], tgt)
    gc, gf = b.grad_views()
    nk.mlp_backward(b.fine, act_f, nk.raw2outputs_backward(raw_f, z_all, rb, opts, nklib.STREAM_NOISE_FINE, df), gf)
    nk.mlp_backward(b.coarse, act_c, nk.raw2outputs_backward(raw_c, z_c, rb, opts, nklib.STREAM_NOISE_COARSE, dc), gc)
    assert abs(la - float(loss)) < 1e-5 * abs(la)
    assert float(a.grad.norm()) > 0
    assert float((a.grad - b.grad).norm() / a.grad.norm()) < 1e-5


def test_split_backward_equals_single_call(nk, dev, teacher):
    """The data-parallel step's two backward calls (fine, then coarse: an all-reduce per network can
    start in between) give the gradient of the single call: bit for bit at a size where the
    networks' products run in separate launches anyway, to summation order where they are shared."""
    from nerf_kinematics_amd.datasets import RaySampler
    for rays, exact in ((16384, True), (512, False)):
        rb, tgt = RaySampler(teacher, rays, seed=6).next(0)
        a, b = _trainer(nk, dev, rays=rays, seed=4), _trainer(nk, dev, rays=rays, seed=4)
        b.overlap_allreduce = True
        a.step(rb, tgt); b.step(rb, tgt)
        if exact:
            assert torch.equal(a.grad, b.grad) and torch.equal(a.params, b.params)
        else:
            assert float((a.grad - b.grad).norm() / a.grad.norm()) < 1e-5


def test_gradient_is_bit_reproducible(nk, dev, teacher):
    """Weight gradients are summed in a fixed order (K-slices through scratch, then one reduce
    kernel): the same step twice gives the same bits, at a shared-launch and at a per-network size."""
    from nerf_kinematics_amd.datasets import RaySampler
    for rays in (256, 8192):
        rb, tgt = RaySampler(teacher, rays, seed=5).next(0)
        a, b = _trainer(nk, dev, rays=rays, seed=4), _trainer(nk, dev, rays=rays, seed=4)
        a.step(rb, tgt); b.step(rb, tgt)
        assert float(a.grad.abs().sum()) > 0
        assert torch.equal(a.grad, b.grad)
        assert torch.equal(a.params, b.params)


def test_checkpoint_resume_continues_the_run(nk, dev, teacher, tmp_path):
    """save -> load into a fresh trainer -> the next steps follow the original run."""
    from nerf_kinematics_amd import checkpoint as ckpt
    from nerf_kinematics_amd.datasets import RaySampler
    a = _trainer(nk, dev, rays=512)
    smp = RaySampler(teacher, 512, seed=3)
    for i in range(5):
        a.step(*smp.next(i))
    path = str(tmp_path / ckpt.checkpoint_name(a.iteration))
    ckpt.save_checkpoint(path, a.export_state(0.1, 10.0), nk.ARCH_LEGO_CKPT, nk.ARCH_LEGO_CKPT)
    b = _trainer(nk, dev, rays=512, seed=999)           # different initial weights: everything must come from the file
    b.import_state(ckpt.state_from_checkpoint(ckpt.load_checkpoint(path), nk.ARCH_LEGO_CKPT, nk.ARCH_LEGO_CKPT))
    b.opts.seed = a.opts.seed
    assert b.iteration == 5 and b.opt_steps == 5
    # the state is restored exactly ...
    assert torch.equal(a.params, b.params) and torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)
    # ... and, gradients being summed in a fixed order, the resumed run IS the original run: same
    # bits in parameters and both Adam moments after three more steps (a wrong step count, learning
    # rate, seed or moment on resume would show in every parameter)
    for i in range(5, 8):
        rb, tgt = smp.next(i)
        la = float(a.step(rb, tgt)); lb = float(b.step(rb, tgt))
        assert abs(la - lb) <= 1e-6 * abs(la)            # (the loss scalar itself is an atomic sum over blocks)
    assert a.opt_steps == b.opt_steps == 8
    assert torch.equal(a.params, b.params)
    assert torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)


def test_reference_checkpoint_weights_render_through_trainer(nk, dev, golden_dir):
    """The reference's trained lego weights, imported as a checkpoint state, render the probe view
    with the statistics recorded for the oracle (opaque centre, empty corners, mean colour)."""
    from nerf_kinematics_amd import checkpoint as ckpt, datasets
    w = np.load(os.path.join(golden_dir, "lego_lowres3_weights.npz"))
    tr = _trainer(nk, dev, fine=128)
    tr.import_state(ckpt.TrainState(int(w["iter"]), torch.from_numpy(w["coarse"]), torch.from_numpy(w["fine"]), [], [],
                                    0, float(w["lr"])))
    assert tr.iteration == 110001
    focal = 0.5 * 100 / math.tan(0.5 * 0.6911112070083618)
    sc = datasets.Scene(None, datasets.spherical_pose(45.0, -30.0, 4.0)[None], 100, 100, focal, focal, 50.0, 50.0, 2.0, 6.0)
    out = tr.render(sc.rays(0, dev))
    rgb = out["rgb_fine"].view(100, 100, 3)
    acc = out["acc_fine"].view(100, 100)
    with open(os.path.join(golden_dir, "lego_probe_stats.json")) as fh:
        stats = json.load(fh)
    np.testing.assert_allclose(rgb.mean((0, 1)).cpu().numpy(), np.array(stats["mean_rgb"]), atol=2e-3)
    assert float(acc[45:55, 45:55].mean()) > 0.99 and float(acc[:10, :10].mean()) < 1e-3


def test_run_nerf_cli_trains_saves_resumes_and_renders(nk, tmp_path):
    """`python run_nerf.py --config ...` end to end on the synthetic scene, in child processes."""
    import yaml
    with open(os.path.join(ROOT, "tests", "golden", "lego-lowres3_config.yml")) as fh:
        cfg = yaml.safe_load(fh)
    cfg["dataset"] = {"type": "teacher", "near": 2, "far": 6, "train_views": 6, "val_views": 2, "resolution": 48}
    cfg["experiment"].update(id="cli", logdir=str(tmp_path), train_iters=40, print_every=10, validate_every=20, save_every=20)
    p = tmp_path / "exp.yml"
    p.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, PYTHONPATH=ROOT)
    run = lambda *extra: subprocess.run([sys.executable, os.path.join(ROOT, "run_nerf.py"), "--config", str(p), *extra],
                                        env=env, capture_output=True, text=True, timeout=300)
    r = run()
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = tmp_path / "cli"
    assert (out / "config.yml").is_file() and (out / "checkpoint00020.ckpt").is_file() and (out / "checkpoint00039.ckpt").is_file()
    recs = [json.loads(l) for l in (out / "log.jsonl").read_text().splitlines()]
    train = [r_["train"] for r_ in recs if "train" in r_]
    val = [r_["validation"] for r_ in recs if "validation" in r_]
    assert [t["iter"] for t in train] == [0, 10, 20, 30, 39] and len(val) == 2
    assert train[-1]["loss"] < train[0]["loss"] and all(math.isfinite(v["psnr_8bit"]) for v in val)
    ck = torch.load(out / "checkpoint00039.ckpt", map_location="cpu", weights_only=True)
    assert ck["iter"] == 39 and float(ck["optimizer_state_dict"]["state"][0]["step"]) == 40.0
    # resume for 10 more iterations
    r = run("--load-checkpoint", str(out / "checkpoint00039.ckpt"), "--iters", "50")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "resumed" in r.stdout and (out / "checkpoint00049.ckpt").is_file()
    # render the held-out views and a pose path
    r = run("--load-checkpoint", str(out / "checkpoint00049.ckpt"), "--render-only")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert len(list((out / "render").glob("*.png"))) == 2
    path = tmp_path / "video.json"
    from nerf_kinematics_amd import datasets
    path.write_text(json.dumps({"frames": [{"transform_matrix": datasets.spherical_pose(10.0 * k, -30, 4).tolist()}
                                           for k in range(3)] + [{"transform_matrix_start": datasets.spherical_pose(90, -20, 4).tolist()}]}))
    r = run("--load-checkpoint", str(out / "checkpoint00049.ckpt"), "--render-only", "--render-poses", str(path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rendered 4 frames" in r.stdout
    # the same run with the iteration replayed as a HIP graph writes the same checkpoint
    r = run("--graph", "--logdir", str(tmp_path / "g"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ckg = torch.load(tmp_path / "g" / "cli" / "checkpoint00039.ckpt", map_location="cpu", weights_only=True)
    for k, v in ck["model_fine_state_dict"].items():
        assert torch.equal(v, ckg["model_fine_state_dict"][k]), k
    assert torch.equal(ck["optimizer_state_dict"]["state"][3]["exp_avg_sq"], ckg["optimizer_state_dict"]["state"][3]["exp_avg_sq"])


def test_run_nerf_cli_two_ranks(nk, tmp_path):
    """Two ranks of the command line sharing the GPU (gloo): every rank draws its own rays, the
    gradient is all-reduced, validation and --render-only renders are sharded by rows over both
    ranks; rank 0 writes the log, the checkpoints and the frames."""
    import socket
    import yaml
    with open(os.path.join(ROOT, "tests", "golden", "lego-lowres3_config.yml")) as fh:
        cfg = yaml.safe_load(fh)
    cfg["dataset"] = {"type": "teacher", "near": 2, "far": 6, "train_views": 6, "val_views": 2, "resolution": 48}
    cfg["experiment"].update(id="two", logdir=str(tmp_path), train_iters=30, print_every=10, validate_every=15, save_every=100)
    cfg["nerf"]["train"]["num_random_rays"] = 512
    p = tmp_path / "exp.yml"
    p.write_text(yaml.safe_dump(cfg))

    def run(*extra):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        procs = []
        for rank in range(2):
            # a rank still running after 140 s dumps its stacks and exits: a stall names its collective
            env = dict(os.environ, PYTHONPATH=ROOT, NK_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", NK_STALL_DUMP_SECONDS="140")
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "run_nerf.py"), "--config", str(p), *extra],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = [q.communicate(timeout=160)[0] for q in procs]
        assert all(q.returncode == 0 for q in procs), outs[0][-1500:] + outs[1][-1500:]
        return outs

    outs = run()
    out = tmp_path / "two"
    recs = [json.loads(l) for l in (out / "log.jsonl").read_text().splitlines()]
    train = [r_["train"] for r_ in recs if "train" in r_]
    val = [r_["validation"] for r_ in recs if "validation" in r_]
    assert [t["iter"] for t in train] == [0, 10, 20, 29] and [v["iter"] for v in val] == [15, 29]
    assert train[-1]["loss"] < train[0]["loss"] and all(math.isfinite(v["psnr_8bit"]) and v["psnr"] > 5 for v in val)
    assert "[TRAIN]" in outs[0] and "[TRAIN]" not in outs[1]
    # once per run: where the multi-rank step's time goes (all-reduce time, library, settings)
    comm = [r_["comm"] for r_ in recs if "comm" in r_]
    assert len(comm) == 1 and comm[0]["world"] == 2 and comm[0]["steps"] >= 5 and comm[0]["allreduce_ms"] > 0
    assert comm[0]["gradient_bytes"] > 0 and "NCCL_ALGO" in comm[0] and "[COMM]" in outs[0]
    ck = out / "checkpoint00029.ckpt"
    assert ck.is_file()
    outs = run("--load-checkpoint", str(ck), "--render-only")
    assert "rendered 2 frames" in outs[0] and len(list((out / "render").glob("*.png"))) == 2
    # the frames the two ranks rendered together equal a single process's
    frames2 = {f.name: f.read_bytes() for f in (out / "render").glob("*.png")}
    for f in (out / "render").glob("*.png"):
        f.unlink()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_nerf.py"), "--config", str(p), "--load-checkpoint", str(ck),
                        "--render-only"], env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert {f.name: f.read_bytes() for f in (out / "render").glob("*.png")} == frames2


def test_wheel_capture_end_to_end(nk, tmp_path):
    """The reference's own robot capture through the whole chain: poses.txt -> transforms.json
    (converter), robot loader (8x down-sampling, every 8th frame held out, NDC rays), the
    wheel experiment shape (4x128, L = 6, 64 + 64 samples) trained for 300 iterations through
    the command line.  The loss on the photographs must fall well below its starting value."""
    import shutil
    import yaml
    cap = tmp_path / "capture"
    shutil.copytree(os.path.join(ROOT, "tests", "golden", "wheel_real_robot"), cap)
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "nerf_kinematics_amd.poses", "--poses", str(cap / "poses.txt"),
                        "--image_folder", str(cap / "images_robot"), "--recenter", "--output", str(cap / "transforms.json")],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    with open(os.path.join(ROOT, "configs", "wheel_robot.yml")) as fh:
        cfg = yaml.safe_load(fh)
    cfg["dataset"]["basedir"] = str(cap)
    cfg["experiment"].update(id="wheel", logdir=str(tmp_path), train_iters=300, print_every=50, validate_every=150,
                             save_every=1000)
    p = tmp_path / "wheel.yml"
    p.write_text(yaml.safe_dump(cfg))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_nerf.py"), "--config", str(p)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    recs = [json.loads(l) for l in (tmp_path / "wheel" / "log.jsonl").read_text().splitlines()]
    train = [x["train"] for x in recs if "train" in x]
    val = [x["validation"] for x in recs if "validation" in x]
    print("wheel losses", [round(t["loss"], 4) for t in train], "val", [(v["view"], round(v["psnr"], 2)) for v in val])
    assert all(math.isfinite(t["loss"]) for t in train)
    assert train[-1]["loss"] < 0.5 * train[0]["loss"]
    assert val and all(v["view"] in (0, 8) for v in val)          # the held-out frames
    assert (tmp_path / "wheel" / "checkpoint00299.ckpt").is_file()


@pytest.mark.parametrize("fmt", ["bf16", "fp8bf8"])
def test_graph_step_equals_eager_step(nk, dev, teacher, fmt):
    """GraphedStep (the iteration captured into a HIP graph, per-iteration scalars read from a
    device-resident nk_step_state) against Trainer.step (scalars by value, one launch per kernel):
    same rays, same loss, bit-identical gradient, parameters and Adam moments over six iterations;
    and the eager issue of the device-state launch sequence gives the same again."""
    from nerf_kinematics_amd.datasets import RaySampler
    from nerf_kinematics_amd.trainer import GraphedStep
    ref = _trainer(nk, dev, rays=512, seed=9)
    smp_ref = RaySampler(teacher, 512, seed=3)
    runs = []
    for capture in (True, False):
        tr = _trainer(nk, dev, rays=512, seed=9)
        runs.append((tr, GraphedStep(tr, RaySampler(teacher, 512, seed=3), capture=capture)))
    assert runs[0][1].graph is not None and runs[1][1].graph is None
    for i in range(6):
        rb, tgt = smp_ref.next(i)
        l_ref = float(ref.step(rb, tgt))
        for tr, gs in runs:
            l = float(gs.step())
            assert torch.equal(gs.smp.ray_batch, rb) and torch.equal(gs.smp.target, tgt)
            assert abs(l - l_ref) <= 1e-6 * abs(l_ref)
            assert torch.equal(tr.grad, ref.grad)
            assert torch.equal(tr.params, ref.params)
            assert torch.equal(tr.exp_avg, ref.exp_avg) and torch.equal(tr.exp_avg_sq, ref.exp_avg_sq)
            assert tr.iteration == ref.iteration and tr.opt_steps == ref.opt_steps
    assert float(ref.grad.abs().sum()) > 0
    # the graph-trained networks render (weight images were re-packed inside the graph)
    out = runs[0][0].render(rb[:64].contiguous())
    out_ref = ref.render(rb[:64].contiguous())
    assert torch.equal(out["rgb_fine"], out_ref["rgb_fine"])


def test_device_state_sampler_ignores_a_view_outside_the_tables(nk, dev, teacher):
    """nk_sample_training_rays_dev reads the view index from device memory when it runs; an index
    outside the camera table / image stack (a step state that was never set) must not be followed."""
    import ctypes as C
    from nerf_kinematics_amd import lib as nklib
    from nerf_kinematics_amd.datasets import RaySampler
    from nerf_kinematics_amd.trainer import GraphedStep
    tr = _trainer(nk, dev, rays=256, seed=2)
    gs = GraphedStep(tr, RaySampler(teacher, 256, seed=3), capture=False)
    L = nk.load()
    st = torch.cuda.current_stream(dev).cuda_stream
    rb = torch.full((256, 11), float("nan"), device=dev)
    tgt = torch.full((256, 3), float("nan"), device=dev)
    for view, drawn in ((gs.n_views, False), (-1, False), (2 ** 30, False), (gs.n_views - 1, True)):
        state = nklib.NkStepState(render_seed=1, pixel_seed=2, pixel_counter=0, view=view)
        sd = torch.zeros(C.sizeof(nklib.NkStepState), dtype=torch.uint8, device=dev)
        nklib.check(L.nk_set_step_state(state, sd.data_ptr(), st))
        nklib.check(L.nk_sample_training_rays_dev(gs.cams.data_ptr(), gs.n_views, teacher.images.data_ptr(), gs.image_stride,
                                                  gs.channels, gs.white, sd.data_ptr(), 256, rb.data_ptr(), tgt.data_ptr(),
                                                  None, st))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(rb).all()) == drawn and bool(torch.isfinite(tgt).all()) == drawn
