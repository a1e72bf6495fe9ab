"""CPU tests of the drop-in boundary: the shared library loads without a GPU, exports every symbol
include/nerf_kinematics.h declares, and its host-only entry points (sizes, plan validation)
behave.  No compute call is made here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "nerf_kinematics.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nk_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from nerf_kinematics_amd import lib
    L = lib.load()
    syms = declared_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(L, s), f"{s} declared in the header but not exported"
        assert s in lib.SIGNATURES, f"{s} has no ctypes prototype"
    assert L.nk_version() == lib.ABI_VERSION == 103


def test_parameter_counts_match_the_reference_checkpoints():
    import nerf_kinematics_amd as nk
    L = nk.load()
    assert L.nk_num_params(nk.ARCH_LEGO_CKPT.to_c()) == 84548 == nk.ARCH_LEGO_CKPT.num_params()
    assert L.nk_num_params(nk.ARCH_WHEEL_CKPT.to_c()) == 81476 == nk.ARCH_WHEEL_CKPT.num_params()
    assert L.nk_num_params(nk.ARCH_RUN_NERF.to_c()) == 595844 == nk.ARCH_RUN_NERF.num_params()
    # fragment streams: forward 1 KiB per (tile, k-step) + one bias block per tile, then transposed,
    # then the forward stream again in the operand order of the 16x16x32 inference kernel
    assert L.nk_packed_bytes(nk.ARCH_RUN_NERF.to_c()) == (1262 + 1100 + 1262) * 1024
    # without view directions: trunk + one four-row layer; the transposed stream loses the direction branch
    novd = nk.Arch(use_viewdirs=False)
    assert L.nk_num_params(novd.to_c()) == novd.num_params() == 595844 - (283 * 128 + 128) - (256 * 256 + 256) - 257 - 387 + 4 * 257
    assert L.nk_packed_bytes(novd.to_c()) == (1262 + 1100 - 4 - 8 * 8 + 1262) * 1024


def test_unsupported_shapes_are_rejected_with_a_message():
    import nerf_kinematics_amd as nk
    from nerf_kinematics_amd import lib
    L = nk.load()
    bad = nk.Arch(depth=8, width=192, skip=4, dir_width=96)
    assert L.nk_num_params(bad.to_c()) == -2
    assert b"unsupported widths" in L.nk_last_error()
    with pytest.raises(lib.NerfKinError):
        lib.check(L.nk_num_params(bad.to_c()))
    assert L.nk_num_params(nk.Arch(depth=9).to_c()) == -2
    assert L.nk_num_params(nk.Arch(n_freq_xyz=11).to_c()) == -2
    # null pointers come back as NK_ERR_ARG, never a crash
    assert L.nk_pack_weights(nk.ARCH_RUN_NERF.to_c(), None, None, None) == -1
    with pytest.raises(ValueError):
        lib.check(-1)


def test_workspace_sizes_scale_with_rays():
    import nerf_kinematics_amd as nk
    L = nk.load()
    o = nk.make_opts(64, 128)
    a = nk.ARCH_RUN_NERF.to_c()
    inf1 = L.nk_render_workspace_bytes(a, a, o, 1024, 0)
    tr1 = L.nk_render_workspace_bytes(a, a, o, 1024, 1)
    tr2 = L.nk_render_workspace_bytes(a, a, o, 2048, 1)
    tr3 = L.nk_render_workspace_bytes(a, a, o, 3072, 1)
    # linear in the rays on top of a fixed part (the partial-sum scratch of the weight-gradient launches)
    assert 0 < inf1 < tr1 < tr2 < tr3 and abs((tr3 - tr2) / (tr2 - tr1) - 1) < 0.01
    # activations: 2528 bf16 per point-evaluation for the 8x256 network, plus one ReLU-mask bit per
    # hidden activation (10 layers x 16 bytes per lane = 32 bytes per point and layer)
    assert L.nk_act_bytes(a, 1024, 192) == 1024 * 192 * (2528 * 2 + 10 * 32)
    assert L.nk_render_workspace_bytes(a, None, o, 1024, 0) == -1      # fine samples without a fine network


def test_config_to_arch_from_reference_yaml(golden_dir):
    """models.coarse of the reference's config.yml + the checkpoint's own keys -> the trained shape."""
    import yaml
    import nerf_kinematics_amd as nk
    cfg = yaml.safe_load(open(os.path.join(golden_dir, "lego-lowres3_config.yml")))
    keys = {f"layers_xyz.{i}.weight": None for i in range(3)}
    a = nk.Arch.from_config(cfg["models"]["coarse"], keys)
    assert a == nk.ARCH_LEGO_CKPT
