"""nk_render_rays_fwd_noalloc (render_rays for callers that cannot allocate: its stage buffers live
in a device array inside the library image) against nk_render_rays_fwd, both through the C ABI.

Same kernels on the same inputs, so every output must be bit-identical.  Two calls in a row, the
second on fewer rays (the array keeps the first call's contents), then the size limit.
"""
import math
import os

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu

NAMES = ("rgb_fine", "disp_fine", "acc_fine", "depth_fine", "rgb_coarse", "disp_coarse", "acc_coarse", "depth_coarse")


def _outputs(lib_mod, R, dev):
    bufs = {n: torch.full((R, 3) if n.startswith("rgb") else (R,), float("nan"), dtype=torch.float32, device=dev)
            for n in NAMES}
    return bufs, lib_mod.NkRenderOut(*[bufs[n].data_ptr() for n in NAMES])


def test_noalloc_render_equals_workspace_render(golden_dir):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU and the HIP library; there is no CPU fallback")
    import nerf_kinematics_amd as nk
    from nerf_kinematics_amd import lib as nklib
    L = nk.load()
    dev = torch.device("cuda:0")
    w = np.load(os.path.join(golden_dir, "lego_lowres3_weights.npz"))
    coarse = nk.NeRFModel(nk.ARCH_LEGO_CKPT, torch.from_numpy(w["coarse"]).to(dev))
    fine = nk.NeRFModel(nk.ARCH_LEGO_CKPT, torch.from_numpy(w["fine"]).to(dev))
    coarse.ensure_packed(); fine.ensure_packed()
    H = W = 48
    focal = 0.5 * W / math.tan(0.5 * 0.6911112070083618)
    rays_all = nk.get_rays(H, W, focal, O.spherical_pose(45.0, -30.0, 4.0), near=2.0, far=6.0, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream

    # second call: fewer rays, other rays, other sample counts, jitter and noise on
    cases = [(rays_all, nk.make_opts(64, 128)),
             (rays_all[517:517 + 301].contiguous(), nk.make_opts(48, 64, perturb=True, radiance_field_noise_std=0.2, seed=9))]
    for rays, opts in cases:
        R = rays.shape[0]
        ws_bytes = nklib.check(L.nk_render_workspace_bytes(coarse._c, fine._c, opts, R, 0))
        ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=dev)
        ws_ptr = (ws.data_ptr() + 255) & ~255
        ref, ref_out = _outputs(nklib, R, dev)
        got, got_out = _outputs(nklib, R, dev)
        nklib.check(L.nk_render_rays_fwd(coarse._c, coarse.packed.data_ptr(), fine._c, fine.packed.data_ptr(),
                                         rays.data_ptr(), R, opts, 0, ws_ptr, ws_bytes, ref_out, st))
        nklib.check(L.nk_render_rays_fwd_noalloc(coarse._c, coarse.packed.data_ptr(), fine._c, fine.packed.data_ptr(),
                                                 rays.data_ptr(), R, opts, got_out, st))
        torch.cuda.synchronize()
        for n in NAMES:
            a, b = ref[n].cpu().numpy(), got[n].cpu().numpy()
            assert np.array_equal(a, b, equal_nan=True), f"{n} differs at R={R}"
        assert torch.isfinite(got["rgb_fine"]).all() and float(got["acc_fine"].max()) > 0.5

    # beyond the built-in buffer: a clear error, nothing launched
    R_big = 40000
    rays_big = rays_all[:1].expand(R_big, 11).contiguous()
    big, big_out = _outputs(nklib, R_big, dev)
    rc = L.nk_render_rays_fwd_noalloc(coarse._c, coarse.packed.data_ptr(), fine._c, fine.packed.data_ptr(),
                                      rays_big.data_ptr(), R_big, nk.make_opts(64, 128), big_out, st)
    assert rc == -4
    assert b"scratch" in L.nk_last_error()
    torch.cuda.synchronize()
    assert torch.isnan(big["rgb_fine"]).all()


def test_noalloc_render_under_graph_capture(golden_dir):
    """The caller the entry point is for: the call captured into a HIP graph (after one plain call
    on the device, which looks the scratch array's address up) and replayed on new rays gives the
    bits of the workspace render."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU and the HIP library; there is no CPU fallback")
    import nerf_kinematics_amd as nk
    from nerf_kinematics_amd import lib as nklib
    L = nk.load()
    dev = torch.device("cuda:0")
    w = np.load(os.path.join(golden_dir, "lego_lowres3_weights.npz"))
    coarse = nk.NeRFModel(nk.ARCH_LEGO_CKPT, torch.from_numpy(w["coarse"]).to(dev))
    fine = nk.NeRFModel(nk.ARCH_LEGO_CKPT, torch.from_numpy(w["fine"]).to(dev))
    H = W = 32
    focal = 0.5 * W / math.tan(0.5 * 0.6911112070083618)
    views = [nk.get_rays(H, W, focal, O.spherical_pose(th, -30.0, 4.0), near=2.0, far=6.0, device=dev) for th in (45.0, 200.0)]
    R = H * W
    opts = nk.make_opts(64, 128)
    rays = views[0].clone()
    got, got_out = _outputs(nklib, R, dev)
    call = lambda st: nklib.check(L.nk_render_rays_fwd_noalloc(coarse._c, coarse.packed.data_ptr(), fine._c,
                                                               fine.packed.data_ptr(), rays.data_ptr(), R, opts, got_out, st))
    call(torch.cuda.current_stream(dev).cuda_stream)           # primes the device (outside capture)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            call(side.cuda_stream)
    torch.cuda.current_stream(dev).wait_stream(side)
    ws_bytes = nklib.check(L.nk_render_workspace_bytes(coarse._c, fine._c, opts, R, 0))
    ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=dev)
    ws_ptr = (ws.data_ptr() + 255) & ~255
    for v in (views[1], views[0]):
        rays.copy_(v)
        for t in got.values():
            t.fill_(float("nan"))
        g.replay()
        ref, ref_out = _outputs(nklib, R, dev)
        nklib.check(L.nk_render_rays_fwd(coarse._c, coarse.packed.data_ptr(), fine._c, fine.packed.data_ptr(),
                                         v.data_ptr(), R, opts, 0, ws_ptr, ws_bytes, ref_out,
                                         torch.cuda.current_stream(dev).cuda_stream))
        torch.cuda.synchronize()
        for n in NAMES:
            assert np.array_equal(ref[n].cpu().numpy(), got[n].cpu().numpy(), equal_nan=True), n
        assert float(got["acc_fine"].max()) > 0.5
