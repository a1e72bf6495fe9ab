"""Host-side mirror of the nerf-pytorch render interface on top of the HIP C ABI.

Function names and argument meaning follow the reference lineage's ``run_nerf.py`` /
``nerf_helpers.py`` vocabulary (``get_rays``, ``ndc_rays``, ``raw2outputs``, ``sample_pdf``,
``render_rays``); the knobs are the ``nerf.train`` / ``nerf.validation`` keys of the experiment
files (/root/reference/trained_models/nerf-pytorch/lego-lowres3/config.yml:46-63).  Every
function launches HIP kernels on the current torch stream; tensors are only used for device
memory.  Bad shapes / dtypes raise ``ValueError`` before any launch.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

from . import lib as _lib
from .model import NeRFModel


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _f32(t: torch.Tensor, name: str, shape=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or t.dtype != torch.float32:
        raise ValueError(f"{name}: expected a float32 tensor")
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a GPU tensor (the render path has no CPU implementation)")
    if shape is not None:
        if t.dim() != len(shape) or any(s is not None and s != d for s, d in zip(shape, t.shape)):
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
    return t.contiguous()


def _ray_rows(ray_batch: torch.Tensor, takes_dirs: bool = False) -> torch.Tensor:
    """Ray-batch rows for the kernels: ``[o(3) d(3) near far viewdir(3)]``, 11 floats.  The reference's
    8-column rows ``[o d near far]`` (``use_viewdirs: False``) are accepted wherever the view direction
    is not needed (``takes_dirs`` false: every stage but a network with a direction branch) and widened
    with zeros; the kernels address rows of 11 floats and never read columns 8..10 in those places."""
    if isinstance(ray_batch, torch.Tensor) and ray_batch.dim() == 2 and ray_batch.shape[1] == 8:
        if takes_dirs:
            raise ValueError("ray_batch: 8-column rows carry no view directions, but the network has a direction "
                             "branch (use_viewdirs); expected shape (None, 11)")
        return torch.nn.functional.pad(_f32(ray_batch, "ray_batch", (None, 8)), (0, 3))
    return _f32(ray_batch, "ray_batch", (None, 11))


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def make_opts(num_coarse: int, num_fine: int, perturb: bool = False, lindisp: bool = False,
              radiance_field_noise_std: float = 0.0, white_background: bool = False,
              seed: int = 0, act_format: str = "bf16", skip_empty_eps: float = 0.0) -> _lib.NkRenderOpts:
    if num_coarse < 2 or num_fine < 0:
        raise ValueError(f"num_coarse={num_coarse} (>= 2) / num_fine={num_fine} (>= 0)")
    if act_format not in _lib.ACT_FORMATS:
        raise ValueError(f"act_format={act_format!r}: expected one of {sorted(_lib.ACT_FORMATS)}")
    if not (skip_empty_eps >= 0.0):
        raise ValueError(f"skip_empty_eps={skip_empty_eps} (>= 0)")
    return _lib.NkRenderOpts(int(num_coarse), int(num_fine), int(bool(perturb)), int(bool(lindisp)),
                             int(bool(white_background)), float(radiance_field_noise_std), int(seed) & 0xFFFFFFFF,
                             _lib.ACT_FORMATS[act_format], None, float(skip_empty_eps), 0)


def dz_scale(R: int) -> float:
    """The power of two ``nk_render_rays_bwd`` multiplies dZ by before rounding it to e5m2 under
    ``act_format="fp8bf8"`` (and divides out of the weight gradient again): 2^(ceil(log2 R) + 6)."""
    return 2.0 ** (math.ceil(math.log2(R)) + 6)


# ---------------------------------------------------------------------------------------------
# stages
# ---------------------------------------------------------------------------------------------
def get_rays(H: int, W: int, focal, c2w: torch.Tensor, *, near: float, far: float, ndc: bool = False,
             ndc_near: float = 1.0, pixels: Optional[torch.Tensor] = None, device=None,
             cx: Optional[float] = None, cy: Optional[float] = None) -> torch.Tensor:
    """Ray batch rows ``[o(3) d(3) near far viewdir(3)]`` for whole image or selected pixels.

    ``focal`` is a float or ``(fx, fy)``; ``pixels`` is an int32 tensor of flat indices row*W+col.
    With ``ndc`` the origins/directions are warped (forward-facing scenes) and view directions
    keep the un-warped normalised direction, as the reference lineage does.
    """
    L = _lib.load()
    fx, fy = (focal, focal) if not isinstance(focal, (tuple, list)) else focal
    m = torch.as_tensor(c2w, dtype=torch.float32).cpu()
    if m.shape[0] < 3 or m.shape[1] != 4:
        raise ValueError(f"c2w: expected [3|4, 4], got {tuple(m.shape)}")
    cam = _lib.NkCamera()
    cam.H, cam.W, cam.fx, cam.fy = int(H), int(W), float(fx), float(fy)
    cam.cx = float(W * 0.5 if cx is None else cx)
    cam.cy = float(H * 0.5 if cy is None else cy)
    for k, v in enumerate(m[:3].reshape(-1).tolist()):
        cam.c2w[k] = v
    cam.near, cam.far, cam.ndc, cam.ndc_near = float(near), float(far), int(bool(ndc)), float(ndc_near)
    if pixels is not None:
        if pixels.dtype != torch.int32 or not pixels.is_cuda:
            raise ValueError("pixels: expected an int32 GPU tensor")
        pixels = pixels.contiguous()
        device = pixels.device
        R = pixels.numel()
    else:
        R = H * W
        if device is None:
            raise ValueError("get_rays: device required when pixels is None")
    out = torch.empty(R, 11, dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        _lib.check(L.nk_get_rays(cam, _ptr(pixels), 0, R, out.data_ptr(), _stream(out.device)))
    return out


def stratified_z(ray_batch: torch.Tensor, opts: _lib.NkRenderOpts) -> torch.Tensor:
    L = _lib.load()
    rb = _ray_rows(ray_batch)
    z = torch.empty(rb.shape[0], opts.num_coarse, dtype=torch.float32, device=rb.device)
    with torch.cuda.device(rb.device):
        _lib.check(L.nk_stratified_z(rb.data_ptr(), rb.shape[0], opts, z.data_ptr(), _stream(rb.device)))
    return z


def run_network(model: NeRFModel, ray_batch: torch.Tensor, z: torch.Tensor,
                act_save: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Embedder + MLP on the points ``o + d z``: raw [R, n, 4] = (rgb_raw, sigma_raw)."""
    L = _lib.load()
    rb = _ray_rows(ray_batch, model.arch.use_viewdirs)
    z = _f32(z, "z", (rb.shape[0], None))
    model.ensure_packed()
    R, n = z.shape
    raw = torch.empty(R, n, 4, dtype=torch.float32, device=rb.device)
    with torch.cuda.device(rb.device):
        _lib.check(L.nk_mlp_forward(model._c, model.packed.data_ptr(), rb.data_ptr(), z.data_ptr(), R, n,
                                    raw.data_ptr(), _ptr(act_save), _stream(rb.device)))
    return raw


def raw2outputs(raw: torch.Tensor, z: torch.Tensor, ray_batch: torch.Tensor, opts: _lib.NkRenderOpts,
                noise_stream: int = _lib.STREAM_NOISE_FINE) -> Dict[str, torch.Tensor]:
    L = _lib.load()
    rb = _ray_rows(ray_batch)
    R = rb.shape[0]
    z = _f32(z, "z", (R, None))
    n = z.shape[1]
    raw = _f32(raw, "raw", (R, n, 4))
    dev = rb.device
    o = {k: torch.empty(R, dtype=torch.float32, device=dev) for k in ("disp", "acc", "depth")}
    o["rgb"] = torch.empty(R, 3, dtype=torch.float32, device=dev)
    o["weights"] = torch.empty(R, n, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.nk_raw2outputs(raw.data_ptr(), z.data_ptr(), rb.data_ptr(), R, n, opts, noise_stream,
                                    o["rgb"].data_ptr(), o["disp"].data_ptr(), o["acc"].data_ptr(),
                                    o["depth"].data_ptr(), o["weights"].data_ptr(), _stream(dev)))
    return o


def raw2outputs_backward(raw, z, ray_batch, opts, noise_stream, d_rgb, d_acc=None, d_depth=None) -> torch.Tensor:
    L = _lib.load()
    rb = _ray_rows(ray_batch)
    R = rb.shape[0]
    z = _f32(z, "z", (R, None))
    n = z.shape[1]
    raw = _f32(raw, "raw", (R, n, 4))
    d_rgb = _f32(d_rgb, "d_rgb", (R, 3))
    d_raw = torch.empty_like(raw)
    with torch.cuda.device(rb.device):
        _lib.check(L.nk_raw2outputs_bwd(raw.data_ptr(), z.data_ptr(), rb.data_ptr(), R, n, opts, noise_stream,
                                        d_rgb.data_ptr(), _ptr(d_acc), _ptr(d_depth), d_raw.data_ptr(),
                                        _stream(rb.device)))
    return d_raw


def sample_pdf(z_coarse: torch.Tensor, weights: torch.Tensor, opts: _lib.NkRenderOpts
               ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Hierarchical samples from the interior coarse weights; returns (sorted z_all, new samples).

    Fast path: ``z_coarse`` ascending along each ray (what ``stratified_z`` produces) -- only the new
    samples are sorted and the two lists merged.  Rays whose coarse depths are not ascending are
    sorted whole, as ``torch.sort(cat[z, z_samples])`` would."""
    L = _lib.load()
    z = _f32(z_coarse, "z_coarse", (None, opts.num_coarse))
    w = _f32(weights, "weights", (z.shape[0], opts.num_coarse))
    R = z.shape[0]
    z_all = torch.empty(R, opts.num_coarse + opts.num_fine, dtype=torch.float32, device=z.device)
    z_s = torch.empty(R, opts.num_fine, dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        _lib.check(L.nk_sample_pdf(z.data_ptr(), w.data_ptr(), R, opts, z_all.data_ptr(), z_s.data_ptr(),
                                   _stream(z.device)))
    return z_all, z_s


# ---------------------------------------------------------------------------------------------
# render_rays
# ---------------------------------------------------------------------------------------------
class _RenderRays(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ray_batch, flat_c, flat_f, coarse, fine, opts, training):
        L = _lib.load()
        dev = ray_batch.device
        R = ray_batch.shape[0]
        Nf = opts.num_fine
        fc = fine._c if fine is not None else None
        ws_bytes = _lib.check(L.nk_render_workspace_bytes(coarse._c, fc, opts, R, int(training)))
        ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=dev)
        ws_ptr = (ws.data_ptr() + 255) & ~255
        mk = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        outs = {"rgb_coarse": mk(R, 3), "disp_coarse": mk(R), "acc_coarse": mk(R), "depth_coarse": mk(R)}
        if Nf > 0:
            outs.update({"rgb_fine": mk(R, 3), "disp_fine": mk(R), "acc_fine": mk(R), "depth_fine": mk(R)})
        o = _lib.NkRenderOut(*[_ptr(outs.get(k)) for k in ("rgb_fine", "disp_fine", "acc_fine", "depth_fine",
                                                            "rgb_coarse", "disp_coarse", "acc_coarse", "depth_coarse")])
        coarse.ensure_packed()
        if fine is not None:
            fine.ensure_packed()
        with torch.cuda.device(dev):
            _lib.check(L.nk_render_rays_fwd(coarse._c, coarse.packed.data_ptr(), fc,
                                            fine.packed.data_ptr() if fine is not None else None,
                                            ray_batch.data_ptr(), R, opts, int(training), ws_ptr, ws_bytes, o,
                                            _stream(dev)))
        ctx.coarse, ctx.fine, ctx.opts, ctx.ws, ctx.ws_ptr, ctx.ws_bytes = coarse, fine, opts, ws, ws_ptr, ws_bytes
        ctx.ray_batch = ray_batch
        ctx.training = training
        if Nf > 0:
            res = (outs["rgb_fine"], outs["disp_fine"], outs["acc_fine"], outs["rgb_coarse"], outs["disp_coarse"],
                   outs["acc_coarse"], outs["depth_fine"], outs["depth_coarse"])
        else:
            res = (outs["rgb_coarse"], outs["disp_coarse"], outs["acc_coarse"], outs["rgb_coarse"],
                   outs["disp_coarse"], outs["acc_coarse"], outs["depth_coarse"], outs["depth_coarse"])
        ctx.mark_non_differentiable(res[1], res[2], res[4], res[5], res[6], res[7])
        return res

    @staticmethod
    def backward(ctx, g_rgb_f, g_disp_f, g_acc_f, g_rgb_c, *_):
        if not ctx.training:
            raise RuntimeError("render_rays: backward needs training=True in the forward call")
        L = _lib.load()
        coarse, fine, opts = ctx.coarse, ctx.fine, ctx.opts
        dev = ctx.ray_batch.device
        R = ctx.ray_batch.shape[0]
        Nf = opts.num_fine
        gc = torch.zeros_like(coarse.flat)
        gf = torch.zeros_like(fine.flat) if fine is not None else None
        if Nf == 0:
            # both result slots alias the coarse map
            g_c = None
            for g in (g_rgb_f, g_rgb_c):
                if g is not None:
                    g_c = g if g_c is None else g_c + g
            g_f = None
        else:
            g_f, g_c = g_rgb_f, g_rgb_c
        g_f = g_f.contiguous() if g_f is not None else None
        g_c = g_c.contiguous() if g_c is not None else None
        with torch.cuda.device(dev):
            _lib.check(L.nk_render_rays_bwd(coarse._c, coarse.packed.data_ptr(), fine._c if fine is not None else None,
                                            fine.packed.data_ptr() if fine is not None else None,
                                            ctx.ray_batch.data_ptr(), R, opts, ctx.ws_ptr, ctx.ws_bytes,
                                            _ptr(g_f), _ptr(g_c), gc.data_ptr(), _ptr(gf), _stream(dev)))
        return None, gc, gf, None, None, None, None


def render_rays(ray_batch: torch.Tensor, coarse: NeRFModel, fine: Optional[NeRFModel] = None, *,
                num_coarse: int = 64, num_fine: int = 128, perturb: bool = False, lindisp: bool = False,
                radiance_field_noise_std: float = 0.0, white_background: bool = False, seed: int = 0,
                training: Optional[bool] = None, act_format: str = "bf16", skip_empty_eps: float = 0.0):
    """Volume-render rays: returns ``(rgb_fine, disp_fine, acc_fine, rgb_coarse, disp_coarse, acc_coarse)``.

    ``ray_batch``: float32 ``[R, 11]`` rows ``[o d near far viewdir]``, or ``[R, 8]`` without the view
    direction when neither network has a direction branch (``Arch.use_viewdirs`` false).
    With ``num_fine == 0`` the fine slots repeat the coarse maps.  Differentiable w.r.t.
    ``coarse.flat`` and ``fine.flat`` (through the rgb maps) when they require grad.

    Two experiments beyond the reference's behaviour, both off by default: ``act_format`` ("fp8",
    "fp8bf8") saves the tensors backward needs in 8-bit formats (training calls), and
    ``skip_empty_eps`` > 0 skips the fine network where the coarse pass found nothing (inference calls).
    """
    if num_fine > 0 and fine is None:
        raise ValueError("render_rays: num_fine > 0 needs a fine model")
    if num_fine == 0:
        fine = None
    rb = _ray_rows(ray_batch, coarse.arch.use_viewdirs or (fine is not None and fine.arch.use_viewdirs))
    opts = make_opts(num_coarse, num_fine, perturb, lindisp, radiance_field_noise_std, white_background, seed,
                     act_format=act_format, skip_empty_eps=skip_empty_eps)
    if training is None:
        training = torch.is_grad_enabled() and (coarse.flat.requires_grad or (fine is not None and fine.flat.requires_grad))
    flat_f = fine.flat if fine is not None else None
    res = _RenderRays.apply(rb, coarse.flat, flat_f, coarse, fine, opts, bool(training))
    return res[:6]


# ---------------------------------------------------------------------------------------------
# MLP backward, loss and optimiser (training-side stages)
# ---------------------------------------------------------------------------------------------
def alloc_activations(model: NeRFModel, R: int, n: int) -> torch.Tensor:
    """Buffer ``run_network(..., act_save=...)`` fills for ``mlp_backward``."""
    L = _lib.load()
    nbytes = _lib.check(L.nk_act_bytes(model._c, R, n))
    return torch.empty(nbytes, dtype=torch.uint8, device=model.flat.device)


def mlp_backward(model: NeRFModel, act_save: torch.Tensor, d_raw: torch.Tensor,
                 grad: Optional[torch.Tensor] = None) -> torch.Tensor:
    """grad (flat, fp32) += dL/dparams given dL/draw [R, n, 4] and the saved activations."""
    L = _lib.load()
    d_raw = _f32(d_raw, "d_raw", (None, None, 4))
    R, n = d_raw.shape[0], d_raw.shape[1]
    dev = d_raw.device
    if grad is None:
        grad = torch.zeros_like(model.flat)
    dz = torch.empty(_lib.check(L.nk_dz_bytes(model._c, R, n)), dtype=torch.uint8, device=dev)
    model.ensure_packed()
    with torch.cuda.device(dev):
        _lib.check(L.nk_mlp_backward(model._c, model.packed.data_ptr(), act_save.data_ptr(), d_raw.data_ptr(), R, n,
                                     dz.data_ptr(), grad.data_ptr(), _stream(dev)))
    return grad


def mse_loss(rgb_coarse: Optional[torch.Tensor], rgb_fine: Optional[torch.Tensor], target: torch.Tensor,
             grad_scale: float = 1.0):
    """img2mse on both maps; returns (loss[1], d_rgb_coarse, d_rgb_fine)."""
    L = _lib.load()
    target = _f32(target, "target", (None, 3))
    R, dev = target.shape[0], target.device
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    dc = torch.empty_like(target) if rgb_coarse is not None else None
    df = torch.empty_like(target) if rgb_fine is not None else None
    with torch.cuda.device(dev):
        _lib.check(L.nk_mse_loss(_ptr(rgb_coarse), _ptr(rgb_fine), target.data_ptr(), R, float(grad_scale),
                                 loss.data_ptr(), _ptr(dc), _ptr(df), _stream(dev)))
    return loss, dc, df


def adam_step(params: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step: int,
              lr: float, betas=(0.9, 0.999), eps: float = 1e-8, grad_scale: float = 1.0) -> None:
    """torch.optim.Adam's update on flat fp32 buffers, in place."""
    L = _lib.load()
    for name, t in (("params", params), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        _f32(t, name, (params.numel(),))
        if not t.is_contiguous():
            raise ValueError(f"{name}: must be contiguous")
    with torch.cuda.device(params.device):
        _lib.check(L.nk_adam_step(params.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                  params.numel(), int(step), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                  float(grad_scale), _stream(params.device)))
