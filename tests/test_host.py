"""CPU tests of the host side around the render path: experiment files, checkpoints in the
reference's layout, metrics, scene loading (no GPU, no compute calls into the library)."""
import json
import math
import os

import numpy as np
import pytest
import torch

from nerf_kinematics_amd import checkpoint as ckpt
from nerf_kinematics_amd import config as config_mod
from nerf_kinematics_amd import datasets, metrics
from nerf_kinematics_amd.model import ARCH_LEGO_CKPT, ARCH_RUN_NERF, ARCH_WHEEL_CKPT, Arch
from nerf_kinematics_amd.trainer import TrainOpts, lr_at
from oracle import nerf_oracle as O


# ---------------------------------------------------------------------------------------------
# experiment files
# ---------------------------------------------------------------------------------------------
def test_reference_configs_load(golden_dir):
    lego = config_mod.load_config(os.path.join(golden_dir, "lego-lowres3_config.yml"))
    assert lego["dataset"]["type"] == "blender" and lego["dataset"]["half_res"] is True
    assert lego["nerf"]["train"]["num_random_rays"] == 1024
    kw = config_mod.render_kwargs(lego, "train")
    assert kw == dict(num_coarse=64, num_fine=64, perturb=True, lindisp=False, radiance_field_noise_std=0.2,
                      white_background=False)
    assert config_mod.render_kwargs(lego, "validation")["perturb"] is False
    o = TrainOpts.from_config(lego)
    assert (o.lr, o.lr_decay, o.lr_decay_factor, o.seed) == (0.005, 250, 0.1, 42)
    wheel = config_mod.load_config(os.path.join(golden_dir, "wheel_robot6_config.yml"))
    assert wheel["dataset"]["type"] == "robot" and wheel["dataset"]["no_ndc"] is False
    assert wheel["dataset"]["downsample_factor"] == 8 and wheel["dataset"]["llffhold"] == 8
    assert wheel["models"]["coarse"]["num_encoding_fn_xyz"] == 6


def test_stored_learning_rate_is_reproduced(golden_dir):
    """lr stored in both reference checkpoints = lr0 * factor ** (iter / (lr_decay * 1000))."""
    for cfg_name, w_name in (("lego-lowres3_config.yml", "lego_lowres3_weights.npz"),
                             ("wheel_robot6_config.yml", "wheel_robot6_weights.npz")):
        o = TrainOpts.from_config(config_mod.load_config(os.path.join(golden_dir, cfg_name)))
        w = np.load(os.path.join(golden_dir, w_name))
        assert lr_at(int(w["iter"]), o) == pytest.approx(float(w["lr"]), rel=1e-12)


@pytest.mark.parametrize("edit,key", [
    (lambda c: c["nerf"].__setitem__("encode_position_fn", "hash_grid"), "encode_position_fn"),
    (lambda c: c["models"]["coarse"].__setitem__("hidden_size", 96), "hidden_size"),
    (lambda c: c["optimizer"].__setitem__("type", "SGD"), "optimizer.type"),
    (lambda c: c["dataset"].__setitem__("type", "llff"), "dataset.type"),
    (lambda c: c["nerf"]["train"].__setitem__("num_coarse", 1), "num_coarse"),
    (lambda c: c["nerf"]["train"].__setitem__("act_format", "fp4"), "act_format"),
    (lambda c: c["nerf"]["validation"].__setitem__("skip_empty_eps", -1.0), "skip_empty_eps"),
    (lambda c: c["nerf"].__setitem__("use_viewdirs", False), "use_viewdirs"),      # the model blocks still say True
    (lambda c: c["models"]["fine"].__setitem__("use_viewdirs", False), "use_viewdirs"),
])
def test_unsupported_config_is_named(golden_dir, tmp_path, edit, key):
    import yaml
    with open(os.path.join(golden_dir, "lego-lowres3_config.yml")) as fh:
        raw = yaml.safe_load(fh)
    edit(raw)
    p = tmp_path / "bad.yml"
    p.write_text(yaml.safe_dump(raw))
    with pytest.raises(ValueError, match=key.split(".")[-1]):
        config_mod.load_config(str(p))


def test_experiment_keys_default_off_and_reach_the_render_options(golden_dir):
    """The two keys the reference's files do not have: absent -> bf16 saved tensors and no skipping
    (render_kwargs then carries nothing the reference's render_rays does not take)."""
    cfg = config_mod.load_config(os.path.join(golden_dir, "lego-lowres3_config.yml"))
    assert TrainOpts.from_config(cfg).act_format == "bf16"
    assert "skip_empty_eps" not in config_mod.render_kwargs(cfg, "validation")
    cfg = config_mod.load_config(os.path.join(golden_dir, "lego-lowres3_config.yml"),
                                 {"nerf": {"train": {"act_format": "fp8bf8"}, "validation": {"skip_empty_eps": 1e-3}}})
    assert TrainOpts.from_config(cfg).act_format == "fp8bf8"
    assert config_mod.render_kwargs(cfg, "validation")["skip_empty_eps"] == 1e-3
    assert "skip_empty_eps" not in config_mod.render_kwargs(cfg, "train")


def test_use_viewdirs_false_builds_networks_without_the_direction_branch(golden_dir, tmp_path):
    """nerf.use_viewdirs and models.*.use_viewdirs (config.yml:30,42,55 of the reference's files) all False:
    either lineage ends in one four-row layer; a checkpoint written for such a pair has those key names,
    loads into a plain torch module tree of the public model class, and round-trips bit-exactly."""
    off = {"nerf": {"use_viewdirs": False}, "models": {"coarse": {"use_viewdirs": False}, "fine": {"use_viewdirs": False}}}
    cfg = config_mod.load_config(os.path.join(golden_dir, "lego-lowres3_config.yml"), off)
    arch = Arch.from_config(cfg["models"]["fine"])
    assert not arch.use_viewdirs and arch.to_c().no_viewdirs == 1 and ARCH_RUN_NERF.to_c().no_viewdirs == 0
    assert [n for n, _ in arch.param_shapes()] == ["layer1", "layers_xyz.0", "layers_xyz.1", "layers_xyz.2", "fc_out"]
    big = Arch.from_config({"type": "NeRF", "hidden_size": 256, "num_layers": 8, "use_viewdirs": False})
    assert [n for n, _ in big.param_shapes()][-2:] == ["pts_linears.7", "output_linear"]
    assert big.num_params() == ARCH_RUN_NERF.num_params() - (283 * 128 + 128) - (256 * 256 + 256) - 257 - (3 * 128 + 3) + 4 * 257
    n = arch.num_params()
    g = torch.Generator().manual_seed(2)
    s = ckpt.TrainState(3, torch.randn(n, generator=g), torch.randn(n, generator=g),
                        [torch.randn(n, generator=g) for _ in range(2)], [torch.rand(n, generator=g) for _ in range(2)],
                        4, 5e-4, 0.01, 20.0)
    path = str(tmp_path / "novd.ckpt")
    ckpt.save_checkpoint(path, s, arch, arch)
    ck = torch.load(path, map_location="cpu", weights_only=True)

    class Flexible(torch.nn.Module):                # FlexibleNeRFModel(use_viewdirs=False): module names only
        def __init__(self):
            super().__init__()
            lin = torch.nn.Linear
            self.layer1 = lin(63, 128)
            self.layers_xyz = torch.nn.ModuleList([lin(128, 128) for _ in range(3)])
            self.fc_out = lin(128, 4)
    m = Flexible()
    m.load_state_dict(ck["model_fine_state_dict"], strict=True)
    assert torch.equal(torch.cat([p.detach().reshape(-1) for p in m.parameters()]), s.fine)
    back = ckpt.state_from_checkpoint(ckpt.load_checkpoint(path), arch, arch)
    assert torch.equal(back.coarse, s.coarse) and torch.equal(back.fine, s.fine)
    assert torch.equal(back.exp_avg[1], s.exp_avg[1]) and torch.equal(back.exp_avg_sq[0], s.exp_avg_sq[0])
    sd = ck["model_coarse_state_dict"]
    assert ckpt.arch_for_state_dict(cfg["models"]["coarse"], sd) == arch
    # a stored network decides by its keys, whatever the model block says
    assert ckpt.arch_for_state_dict(dict(cfg["models"]["coarse"], use_viewdirs=True), sd) == arch


def test_encoding_variants_reach_the_network_description(golden_dir, tmp_path):
    """include_input_* / log_sampling_* of the models block (config.yml:21-24): input widths, the C
    struct's flag bits, and the depth / frequency detection from a stored network."""
    from nerf_kinematics_amd import lib as nklib
    over = {"models": {"coarse": {"include_input_xyz": False, "log_sampling_dir": False},
                       "fine": {"include_input_dir": False, "log_sampling_xyz": False}}}
    cfg = config_mod.load_config(os.path.join(golden_dir, "lego-lowres3_config.yml"), over)
    ac, af = Arch.from_config(cfg["models"]["coarse"]), Arch.from_config(cfg["models"]["fine"])
    assert (ac.in_xyz, ac.in_dir, af.in_xyz, af.in_dir) == (60, 27, 63, 24)
    assert ac.to_c().enc_flags == nklib.NK_ENC_NO_INPUT_XYZ | nklib.NK_ENC_LINEAR_DIR
    assert af.to_c().enc_flags == nklib.NK_ENC_NO_INPUT_DIR | nklib.NK_ENC_LINEAR_XYZ
    assert ARCH_LEGO_CKPT.to_c().enc_flags == 0
    assert dict(ac.param_shapes())["layer1"] == (128, 60) and dict(af.param_shapes())["layers_dir.0"] == (64, 128 + 24)
    assert ac.num_params() == ARCH_LEGO_CKPT.num_params() - 3 * 128 and af.num_params() == ARCH_LEGO_CKPT.num_params() - 3 * 64
    L = nklib.load()
    assert L.nk_num_params(ac.to_c()) == ac.num_params() and L.nk_num_params(af.to_c()) == af.num_params()
    bad = ac.to_c(); bad.enc_flags = 64
    assert L.nk_num_params(bad) == -1 and b"enc_flags" in L.nk_last_error()
    # a stored network: the frequency count follows from the first layer's width and the block's switch
    sd = ckpt.split_flat(af, torch.zeros(af.num_params()))
    assert ckpt.arch_for_state_dict(cfg["models"]["fine"], sd) == af
    sd = ckpt.split_flat(ac, torch.zeros(ac.num_params()))
    assert ckpt.arch_for_state_dict(cfg["models"]["coarse"], sd) == ac


def test_config_round_trip(golden_dir, tmp_path):
    cfg = config_mod.load_config(os.path.join(golden_dir, "wheel_robot6_config.yml"))
    config_mod.dump_config(cfg, str(tmp_path / "config.yml"))
    assert config_mod.load_config(str(tmp_path / "config.yml")) == cfg


# ---------------------------------------------------------------------------------------------
# checkpoints
# ---------------------------------------------------------------------------------------------
def _state(golden_dir, name, arch):
    w = np.load(os.path.join(golden_dir, name))
    g = torch.Generator().manual_seed(0)
    n = arch.num_params()
    m = [torch.randn(n, generator=g) * 1e-3 for _ in range(2)]
    v = [torch.rand(n, generator=g) * 1e-6 for _ in range(2)]
    return ckpt.TrainState(int(w["iter"]), torch.from_numpy(w["coarse"]), torch.from_numpy(w["fine"]), m, v,
                           int(w["adam_step"]), float(w["lr"]), float(w["loss"]), float(w["psnr"]))


@pytest.mark.parametrize("layout,weights,arch", [("lego_ckpt_layout.json", "lego_lowres3_weights.npz", ARCH_LEGO_CKPT),
                                                 ("wheel_ckpt_layout.json", "wheel_robot6_weights.npz", ARCH_WHEEL_CKPT)])
def test_written_checkpoint_has_the_reference_layout(golden_dir, tmp_path, layout, weights, arch):
    """Keys, key ORDER, shapes, dtypes, optimiser bookkeeping of a written file equal those of the
    reference's own checkpoint (fixture: its structure, dumped by tools/make_golden.py)."""
    with open(os.path.join(golden_dir, layout)) as fh:
        ref = json.load(fh)
    path = str(tmp_path / ckpt.checkpoint_name(110000))
    assert os.path.basename(path) == "checkpoint110000.ckpt"
    ckpt.save_checkpoint(path, _state(golden_dir, weights, arch), arch, arch)
    ck = torch.load(path, map_location="cpu", weights_only=True)          # loads with the safe loader
    assert {k: type(v).__name__ for k, v in ck.items()} == ref["top_level"]
    assert list(ck.keys()) == list(ref["top_level"].keys())
    desc = lambda t: [list(t.shape), str(t.dtype)]
    for which in ("model_coarse_state_dict", "model_fine_state_dict"):
        assert list(ck[which].keys()) == list(ref[which].keys())
        assert {k: desc(v) for k, v in ck[which].items()} == ref[which]
    og = ck["optimizer_state_dict"]
    assert {str(i): {k: desc(v) for k, v in e.items()} for i, e in og["state"].items()} == ref["optimizer_state"]
    assert float(og["state"][0]["step"]) == ref["optimizer_step"]
    got_groups = [{k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in g.items()} for g in og["param_groups"]]
    assert got_groups == ref["param_groups"]
    assert ck["iter"] == ref["iter"] and desc(ck["loss"]) == ref["loss"]
    assert ck["psnr"] == ref["psnr"] == pytest.approx(-10 * math.log10(float(ck["loss"])), abs=1e-5)


def test_checkpoint_round_trip_is_bit_exact(golden_dir, tmp_path):
    s = _state(golden_dir, "lego_lowres3_weights.npz", ARCH_LEGO_CKPT)
    path = str(tmp_path / "c.ckpt")
    ckpt.save_checkpoint(path, s, ARCH_LEGO_CKPT, ARCH_LEGO_CKPT)
    ck = ckpt.load_checkpoint(path)
    arch = ckpt.arch_for_state_dict({"type": "FlexibleNeRFModel", "hidden_size": 128, "num_layers": 8},
                                    ck["model_coarse_state_dict"])
    assert arch == ARCH_LEGO_CKPT                       # depth comes from the tensors, not from num_layers
    t = ckpt.state_from_checkpoint(ck, arch, arch)
    assert torch.equal(t.coarse, s.coarse) and torch.equal(t.fine, s.fine)
    for a, b in zip(t.exp_avg + t.exp_avg_sq, s.exp_avg + s.exp_avg_sq):
        assert torch.equal(a, b)
    assert (t.iteration, t.opt_step, t.lr) == (s.iteration, s.opt_step, s.lr)


def test_checkpoint_loads_into_torch_modules_and_adam(golden_dir, tmp_path):
    """What upstream code does with the file: strict load_state_dict + Adam.load_state_dict."""
    arch = ARCH_RUN_NERF
    n = arch.num_params()
    g = torch.Generator().manual_seed(1)
    s = ckpt.TrainState(7, torch.randn(n, generator=g), torch.randn(n, generator=g),
                        [torch.randn(n, generator=g) for _ in range(2)], [torch.rand(n, generator=g) for _ in range(2)],
                        8, 5e-4, 0.01, 20.0)
    path = str(tmp_path / "c.ckpt")
    ckpt.save_checkpoint(path, s, arch, arch)
    ck = torch.load(path, map_location="cpu", weights_only=True)

    class RunNerf(torch.nn.Module):                 # module tree with run_nerf.py's names
        def __init__(self):
            super().__init__()
            lin = torch.nn.Linear
            self.pts_linears = torch.nn.ModuleList([lin(63, 256)] + [lin(256 + (63 if i == 4 else 0), 256) for i in range(7)])
            self.views_linears = torch.nn.ModuleList([lin(283, 128)])
            self.feature_linear, self.alpha_linear, self.rgb_linear = lin(256, 256), lin(256, 1), lin(128, 3)
    mc, mf = RunNerf(), RunNerf()
    mc.load_state_dict(ck["model_coarse_state_dict"], strict=True)
    mf.load_state_dict(ck["model_fine_state_dict"], strict=True)
    opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()), lr=1.0)
    opt.load_state_dict(ck["optimizer_state_dict"])
    assert opt.param_groups[0]["lr"] == 5e-4
    flat = torch.cat([p.detach().reshape(-1) for p in mc.parameters()])
    assert torch.equal(flat, s.coarse)
    first = opt.state[next(iter(mc.parameters()))]
    assert torch.equal(first["exp_avg"].reshape(-1), s.exp_avg[0][:256 * 63]) and float(first["step"]) == 8.0


def test_state_dict_mismatch_raises(golden_dir):
    w = np.load(os.path.join(golden_dir, "lego_lowres3_weights.npz"))
    sd = ckpt.split_flat(ARCH_LEGO_CKPT, torch.from_numpy(w["coarse"]))
    with pytest.raises(ValueError, match="missing"):
        ckpt.join_state_dict(ARCH_RUN_NERF, sd)
    sd["layer1.weight"] = sd["layer1.weight"][:, :39]
    with pytest.raises(ValueError, match="layer1.weight"):
        ckpt.join_state_dict(ARCH_LEGO_CKPT, sd)


# ---------------------------------------------------------------------------------------------
# metrics
# ---------------------------------------------------------------------------------------------
def test_psnr_8bit_formula():
    a = np.zeros((4, 5, 3), np.uint8)
    b = np.full((4, 5, 3), 255, np.uint8)
    assert metrics.psnr_8bit(a, a) == float("inf")
    assert metrics.psnr_8bit(a, b) == pytest.approx(0.0, abs=1e-6)
    c = a.copy(); c[0, 0, 0] = 60                          # mse = 3600 / 60 = 60
    assert metrics.psnr_8bit(a, c) == pytest.approx(20 * math.log10(255 / math.sqrt(60)), rel=1e-6)
    # uint8 inputs must not wrap: 10 - 250 is -240, not 16
    d = np.full((2, 2, 3), 10, np.uint8); e = np.full((2, 2, 3), 250, np.uint8)
    assert metrics.psnr_8bit(d, e) == pytest.approx(20 * math.log10(255 / 240), rel=1e-6)
    with pytest.raises(ValueError):
        metrics.psnr_8bit(a, a[:2])
    assert metrics.mse2psnr(0.0079) == pytest.approx(21.02, abs=0.01)


def test_pose_statistics_match_the_reference_notebook(golden_dir):
    """Numbers printed by the reference's pose analysis for the wheel capture
    (/root/reference/paralaje.ipynb, stored output of its first cell)."""
    with open(os.path.join(golden_dir, "wheel_transforms", "transforms.json")) as fh:
        frames = json.load(fh)["frames"]
    m = metrics.analyze_poses(frames, "wheel (robot)")["metrics"]
    assert m["num_frames"] == 11
    assert m["mean_distance"] == pytest.approx(1.019659, abs=5e-7)
    assert m["spatial_std"] == pytest.approx(0.327081, abs=5e-7)
    assert m["spatial_range"] == pytest.approx(0.974158, abs=5e-7)
    assert m["directional_std"] == pytest.approx(0.13272, abs=5e-6)


# ---------------------------------------------------------------------------------------------
# scenes (host part) and the sampler's host-side draws
# ---------------------------------------------------------------------------------------------
def test_robot_scene_from_converter_output(golden_dir, tmp_path):
    """The wheel capture through the converter's transforms.json, down-sampled 8x as the wheel
    experiment file asks: 160x90, fl_x/fl_y scaled, every 8th frame held out."""
    from nerf_kinematics_amd import poses
    base = os.path.join(golden_dir, "wheel_real_robot")
    docs = poses.convert(os.path.join(base, "poses.txt"), os.path.join(base, "images_robot"), recenter=True, verbose=False)
    path = str(tmp_path / "transforms.json")
    poses.write(docs, path, verbose=False)
    meta = datasets.read_transforms(path)
    assert (meta["w"], meta["h"]) == (1280, 720) and len(meta["frames"]) == 11
    sc = datasets.load_transforms(path, downsample=8, near=0.0, far=1.0, ndc=True, holdout=8, device="cpu")
    assert (sc.H, sc.W) == (90, 160) and tuple(sc.images.shape) == (11, 90, 160, 3) and sc.images.dtype == torch.uint8
    assert sc.fx == pytest.approx(meta["fl_x"] / 8) and sc.fy == pytest.approx(meta["fl_y"] / 8)
    assert (sc.cx, sc.cy) == (80.0, 45.0)
    assert sc.i_test.tolist() == [0, 8] and sc.i_train.tolist() == [1, 2, 3, 4, 5, 6, 7, 9, 10]
    cam = sc.camera(3)
    assert cam.ndc == 1 and cam.H == 90 and abs(cam.c2w[3] - sc.poses[3][0, 3]) < 1e-7
    # area down-sampling keeps the mean colour
    from PIL import Image
    full = np.asarray(Image.open(sc.names[0]).convert("RGB"), dtype=np.float64)
    assert abs(full.mean() - sc.images[0].double().mean().item()) < 0.6


def test_blender_scene_layout(tmp_path):
    """nerf_synthetic layout: three split files, file_path without extension, RGBA PNGs, half_res."""
    from PIL import Image
    rng = np.random.default_rng(0)
    for split, n in (("train", 3), ("val", 4), ("test", 4)):
        os.makedirs(tmp_path / split, exist_ok=True)
        frames = []
        for k in range(n):
            Image.fromarray(rng.integers(0, 255, (8, 8, 4), dtype=np.uint8), "RGBA").save(tmp_path / split / f"r_{k}.png")
            frames.append({"file_path": f"./{split}/r_{k}", "transform_matrix": datasets.spherical_pose(30 * k, -30, 4).tolist()})
        (tmp_path / f"transforms_{split}.json").write_text(json.dumps({"camera_angle_x": 0.6911112070083618, "frames": frames}))
    sc = datasets.load_blender(str(tmp_path), half_res=True, testskip=2, device="cpu")
    assert (sc.H, sc.W) == (4, 4) and tuple(sc.images.shape) == (3 + 2 + 2, 4, 4, 4)
    assert sc.fx == pytest.approx(0.5 * 4 / math.tan(0.5 * 0.6911112070083618))
    assert sc.i_train.tolist() == [0, 1, 2] and sc.i_val.tolist() == [3, 4] and sc.i_test.tolist() == [5, 6]
    sc.white_background = True
    t = sc.target(0)
    px = sc.images[0].reshape(-1, 4).float() / 255
    assert torch.allclose(t, px[:, :3] * px[:, 3:] + 1 - px[:, 3:])


def test_spherical_pose_matches_oracle():
    for th, ph, r in ((45.0, -30.0, 4.0), (200.0, -12.5, 3.0)):
        np.testing.assert_allclose(datasets.spherical_pose(th, ph, r), O.spherical_pose(th, ph, r).numpy(), atol=1e-6)


def test_host_rng_matches_oracle():
    idx = [0, 1, 77, 2 ** 32 - 1, 2 ** 32, 2 ** 40 + 12345]
    for seed, stream in ((42, 4), (0xDEADBEEF, 0), (7, 3)):
        ref = O.rand_u32(seed, stream, np.array(idx, dtype=np.uint64))
        assert [datasets.rand_u32(seed, stream, i) for i in idx] == [int(v) for v in ref]
