"""CPU oracle for the classic-NeRF ``render_rays`` path.  TEST INFRASTRUCTURE ONLY.

This module is the checker the HIP path is compared against.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it; the
product package never does.

What it restates
----------------
The reference project trains and renders with a YAML-configured nerf-pytorch fork that lives in
an un-vendored git submodule (``/root/reference/.gitmodules:10-12``; the directory is empty and
the pinned commit is not recoverable).  The algorithm restated here is therefore the published
NeRF pipeline (Mildenhall et al. 2020, sections 4, 5.1-5.3 and appendix C for NDC rays) in the
form the public PyTorch re-implementations use, written from the mathematics, function by
function:

  get_rays / ndc_rays      pinhole rays, Blender camera convention / forward-facing NDC warp
  positional_encoding      gamma(x) = [x, sin(2^k x), cos(2^k x)]_k   (no pi factor)
  NeRFNet                  the two MLP lineages (see ``Arch``)
  raw2outputs              alpha compositing
  sample_pdf               inverse-CDF hierarchical sampling
  render_rays              coarse -> sample_pdf -> sort -> fine

Knobs and defaults follow the reference's own experiment files:
``/root/reference/trained_models/nerf-pytorch/lego-lowres3/config.yml:18-63`` (model block,
``nerf.train`` / ``nerf.validation`` blocks) and the checkpoint layout of
``/root/reference/trained_models/nerf-pytorch/*/checkpoint110000.ckpt`` (state-dict key names and
shapes: ``layer1``, ``layers_xyz.N``, ``fc_feat``, ``fc_alpha``, ``layers_dir.0``, ``fc_rgb``).

How it is pinned
----------------
The reference holds no tests, golden vectors or fixtures for this path.  The pin is behavioural:
only one combination of conventions (no ReLU after ``layer1``, ReLU on ``fc_feat``,
``cat[feat, gamma(d)]``, Blender ray convention, no pi in the encoding) renders the vendored lego
checkpoint as a coherent object: opaque centre, empty corners, mean colour about
(0.18, 0.15, 0.10).  ``tests/test_oracle.py`` asserts exactly that on the weights extracted to
``tests/golden/``.  ``ndc_rays`` and the 8x256 ``run_nerf`` architecture have no such pin:
parity unpinned for those (they follow the paper).  The reference's one rendered nerf-pytorch image
of its robot capture (``/root/reference/results/wheel/nerf-pytorch/imgs/rendered/TestNERF_0_original_gen.png``
with its target under ``imgs/reals/``) pins the loader's held-out view and image reduction
(``tests/test_wheel_pin.py``) but not this file: a bounded search over camera conventions
(``tools/pin_wheel_search.py``) did not reproduce that render from the vendored wheel weights.

Random numbers
--------------
``perturb`` and ``radiance_field_noise_std`` draw from a counter-based integer hash
(``rand_u32``) so the HIP kernels and this file produce the same streams from a seed.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch

# --------------------------------------------------------------------------------------------
# counter-based RNG shared with the HIP kernels (integer part is bit-exact)
# --------------------------------------------------------------------------------------------
STREAM_STRATIFIED = 0
STREAM_PDF = 1
STREAM_NOISE_COARSE = 2
STREAM_NOISE_FINE = 3
STREAM_PIXELS = 4

_M32 = np.uint64(0xFFFFFFFF)


def _lowbias32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def rand_u32(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """32 random bits for element ``idx`` of ``stream`` under ``seed`` (all arithmetic mod 2^32)."""
    idx = np.asarray(idx, dtype=np.uint64)
    key = _lowbias32(np.uint64((seed & 0xFFFFFFFF) ^ ((stream * 0x85EBCA6B) & 0xFFFFFFFF)))
    lo = idx & _M32
    hi = idx >> np.uint64(32)
    x = (lo * np.uint64(0x9E3779B9) + key) & _M32
    x = _lowbias32(x ^ ((hi * np.uint64(0xC2B2AE35)) & _M32))
    return x.astype(np.uint32)


def rand_uniform(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """U[0,1) with 24 random bits, float32."""
    return ((rand_u32(seed, stream, idx) >> np.uint32(8)).astype(np.float32)
            * np.float32(1.0 / 16777216.0))


def rand_normal(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """N(0,1) by Box-Muller from elements 2*idx and 2*idx+1 of the stream, float32."""
    idx = np.asarray(idx, dtype=np.uint64)
    a = rand_u32(seed, stream, idx * np.uint64(2))
    b = rand_u32(seed, stream, idx * np.uint64(2) + np.uint64(1))
    u1 = ((a >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(1.0 / 16777216.0)
    u2 = (b >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    r = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
    return (r * np.cos(np.float32(2.0 * math.pi) * u2)).astype(np.float32)


# --------------------------------------------------------------------------------------------
# rays
# --------------------------------------------------------------------------------------------
def _fxfy(focal):
    return (focal, focal) if not isinstance(focal, (tuple, list)) else (focal[0], focal[1])


def get_rays(H: int, W: int, focal, c2w: torch.Tensor, cx: Optional[float] = None,
             cy: Optional[float] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Pinhole rays, Blender/OpenGL camera convention (x right, y up, camera looks down -z).

    dir = ((i - W/2)/f, -(j - H/2)/f, -1), rays_d = R @ dir (not normalised), rays_o = t.
    Returns ``rays_o, rays_d`` of shape [H, W, 3]; pixel (row j, column i).  ``focal`` may be a
    pair (fx, fy) and the principal point may be given, for cameras described by
    ``fl_x fl_y cx cy`` (/root/reference/parser_instant_ngp.py:128-145).
    """
    c2w = torch.as_tensor(c2w)
    dt = c2w.dtype
    fx, fy = _fxfy(focal)
    cx = W * 0.5 if cx is None else cx
    cy = H * 0.5 if cy is None else cy
    j, i = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, 3].expand(rays_d.shape)
    return rays_o, rays_d


def ndc_rays(H: int, W: int, focal, near: float, rays_o: torch.Tensor,
             rays_d: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Forward-facing NDC warp (NeRF paper appendix C).  Parity unpinned (no reference fixture).
    ``focal`` may be a pair (fx, fy): the x row of the projection uses fx, the y row fy."""
    fx, fy = _fxfy(focal)
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    o0 = -1.0 / (W / (2.0 * fx)) * rays_o[..., 0] / rays_o[..., 2]
    o1 = -1.0 / (H / (2.0 * fy)) * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1.0 + 2.0 * near / rays_o[..., 2]
    d0 = -1.0 / (W / (2.0 * fx)) * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = -1.0 / (H / (2.0 * fy)) * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2.0 * near / rays_o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


def make_ray_batch(rays_o: torch.Tensor, rays_d: torch.Tensor, near: float, far: float,
                   use_viewdirs: bool = True) -> torch.Tensor:
    """Rows ``[o(3) d(3) near far viewdir(3)]`` (11 floats, or 8 without view directions)."""
    rays_o = rays_o.reshape(-1, 3)
    rays_d = rays_d.reshape(-1, 3)
    cols = [rays_o, rays_d, torch.full_like(rays_o[:, :1], near), torch.full_like(rays_o[:, :1], far)]
    if use_viewdirs:
        cols.append(rays_d / torch.norm(rays_d, dim=-1, keepdim=True))
    return torch.cat(cols, -1)


# --------------------------------------------------------------------------------------------
# positional encoding
# --------------------------------------------------------------------------------------------
def positional_encoding(x: torch.Tensor, num_freqs: int, include_input: bool = True,
                        log_sampling: bool = True) -> torch.Tensor:
    """[x, sin(f0 x), cos(f0 x), sin(f1 x), cos(f1 x), ...]; each term acts on the whole vector.

    Frequencies 2^linspace(0, L-1, L) (``log_sampling_*: true``, config.yml:23-24), no pi factor.
    """
    out = [x] if include_input else []
    if log_sampling:
        freqs = 2.0 ** torch.linspace(0.0, num_freqs - 1, num_freqs, dtype=x.dtype)
    else:
        freqs = torch.linspace(1.0, 2.0 ** (num_freqs - 1), num_freqs, dtype=x.dtype)
    for f in freqs:
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


# --------------------------------------------------------------------------------------------
# MLP
# --------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Arch:
    """Network shape.  Two lineages share one description.

    ``flexible`` (what the vendored checkpoints hold): ``layer1`` (no ReLU) then ``depth-1``
    ReLU layers, ``fc_alpha``, ``fc_feat`` + ReLU, ``layers_dir.0`` on cat[feat, gamma(d)] + ReLU,
    ``fc_rgb``.  Parameter order = state-dict order of the checkpoint.

    ``run_nerf`` (the 8x256 network of the paper): ``pts_linears.0..D-1`` all ReLU, layer
    ``skip+1`` takes cat[gamma(x), h]; ``alpha_linear``; ``feature_linear`` (no ReLU);
    ``views_linears.0`` on cat[feat, gamma(d)] + ReLU; ``rgb_linear``.  Parameter order:
    pts_linears, views_linears, feature_linear, alpha_linear, rgb_linear.

    ``use_viewdirs=False`` (``models.*.use_viewdirs``, config.yml:30,42, and ``nerf.use_viewdirs``, :55):
    either lineage without its direction branch, the trunk followed by one four-row linear layer
    (rgb_raw, sigma_raw) called ``fc_out`` / ``output_linear`` in the public nerf-pytorch sources;
    ray-batch rows then have 8 columns.  Parity unpinned: the reference ships no such network.
    """
    depth: int = 8
    width: int = 256
    skip: int = 4            # -1: no skip connection
    dir_width: int = 128
    n_freq_xyz: int = 10
    n_freq_dir: int = 4
    relu_first: bool = True
    relu_feat: bool = False
    lineage: str = "run_nerf"   # or "flexible"
    use_viewdirs: bool = True
    include_input_xyz: bool = True      # config.yml:21-24: gamma() starts with the raw 3-vector ...
    include_input_dir: bool = True
    log_sampling_xyz: bool = True       # ... frequencies 2^l (True) or linspace(1, 2^(L-1), L) (False)
    log_sampling_dir: bool = True

    @property
    def in_xyz(self) -> int:
        return (3 if self.include_input_xyz else 0) + 6 * self.n_freq_xyz

    @property
    def in_dir(self) -> int:
        return (3 if self.include_input_dir else 0) + 6 * self.n_freq_dir

    def param_shapes(self):
        """[(name, (out, in))] in flat/state-dict order (weight then bias for each)."""
        W, D = self.width, self.depth
        trunk = []
        for i in range(D):
            k = self.in_xyz if i == 0 else (W + self.in_xyz if (self.skip >= 0 and i == self.skip + 1) else W)
            trunk.append((W, k))
        dir_shape = (self.dir_width, W + self.in_dir)
        if self.lineage == "flexible":
            names = [("layer1", trunk[0])]
            names += [(f"layers_xyz.{i - 1}", trunk[i]) for i in range(1, D)]
            if not self.use_viewdirs:
                return names + [("fc_out", (4, W))]
            names += [("layers_dir.0", dir_shape), ("fc_alpha", (1, W)),
                      ("fc_rgb", (3, self.dir_width)), ("fc_feat", (W, W))]
        else:
            names = [(f"pts_linears.{i}", trunk[i]) for i in range(D)]
            if not self.use_viewdirs:
                return names + [("output_linear", (4, W))]
            names += [("views_linears.0", dir_shape), ("feature_linear", (W, W)),
                      ("alpha_linear", (1, W)), ("rgb_linear", (3, self.dir_width))]
        return names

    def num_params(self) -> int:
        return sum(o * i + o for _, (o, i) in self.param_shapes())


ARCH_LEGO_CKPT = Arch(depth=4, width=128, skip=-1, dir_width=64, n_freq_xyz=10, n_freq_dir=4,
                      relu_first=False, relu_feat=True, lineage="flexible")
ARCH_WHEEL_CKPT = Arch(depth=4, width=128, skip=-1, dir_width=64, n_freq_xyz=6, n_freq_dir=4,
                       relu_first=False, relu_feat=True, lineage="flexible")
ARCH_RUN_NERF = Arch()


def _bf16r(x: torch.Tensor) -> torch.Tensor:
    """Round to bfloat16 and back, straight-through for autograd."""
    r = x.detach().to(torch.bfloat16).to(x.dtype)
    return x + (r - x.detach())


class _RoundGradBf16(torch.autograd.Function):
    """Identity whose backward rounds the incoming gradient to bf16: the HIP backward keeps every
    dL/d(pre-activation) as bf16 MFMA operands (fp32 accumulate)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _e4m3r(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float8_e4m3fn).to(torch.float32)


def _e5m2r(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float8_e5m2).to(torch.float32)


class _LinearSaved8(torch.autograd.Function):
    """Linear layer of the bf16-emulating network whose WEIGHT-GRADIENT product sees 8-bit copies of
    its operands, as the HIP path's experimental saved-tensor formats do (no counterpart in the
    reference): the layer input rounded once from fp32 to OCP e4m3 ("fp8", "fp8bf8") and dL/d(output)
    scaled by a power of two, rounded once from fp32 to OCP e5m2 and scaled back ("fp8bf8").  The
    forward product and the gradient passed on to the layer below use bf16 operands as before."""

    @staticmethod
    def forward(ctx, x, w, b, act_format, dz_scale):
        wb = w.to(torch.bfloat16).to(torch.float32)
        ctx.save_for_backward(x, wb)
        ctx.act_format, ctx.dz_scale = act_format, dz_scale
        return x.to(torch.bfloat16).to(torch.float32) @ wb.t() + b

    @staticmethod
    def backward(ctx, g):
        x, wb = ctx.saved_tensors
        g16 = g.to(torch.bfloat16).to(torch.float32)
        gz = _e5m2r(g * ctx.dz_scale) / ctx.dz_scale if ctx.act_format == "fp8bf8" else g16
        lead = x.shape[:-1].numel()
        gz2, x2 = gz.reshape(lead, -1), _e4m3r(x).reshape(lead, -1)
        return g16 @ wb, gz2.t() @ x2, gz2.sum(0), None, None


class NeRFNet:
    """MLP evaluated from a dict of named tensors (``name.weight`` / ``name.bias``).

    ``emulate_bf16``: round weights and every layer input to bf16 (fp32 accumulate, fp32 bias) and,
    in backward, every dL/d(pre-activation); this is where the HIP MFMA path rounds, so tests can
    use a tight tolerance.  ``act_format`` ("fp8", "fp8bf8", with ``dz_scale``) additionally rounds
    the operands of the weight-gradient products as the HIP path's 8-bit saved-tensor formats do.
    """

    def __init__(self, arch: Arch, params: Dict[str, torch.Tensor], emulate_bf16: bool = False,
                 act_format: str = "bf16", dz_scale: float = 1.0):
        self.arch = arch
        self.p = params
        self.emu = emulate_bf16
        if act_format not in ("bf16", "fp8", "fp8bf8"):
            raise ValueError(f"act_format={act_format!r}")
        if act_format != "bf16" and not emulate_bf16:
            raise ValueError("the 8-bit saved-tensor formats are emulated on top of emulate_bf16")
        self.act_format, self.dz_scale = act_format, float(dz_scale)

    # -- flat <-> named ---------------------------------------------------------------------
    @staticmethod
    def from_flat(arch: Arch, flat: torch.Tensor, emulate_bf16: bool = False, act_format: str = "bf16",
                  dz_scale: float = 1.0) -> "NeRFNet":
        p, off = {}, 0
        for name, (o, i) in arch.param_shapes():
            p[name + ".weight"] = flat[off:off + o * i].view(o, i); off += o * i
            p[name + ".bias"] = flat[off:off + o]; off += o
        assert off == flat.numel(), (off, flat.numel())
        return NeRFNet(arch, p, emulate_bf16, act_format, dz_scale)

    @staticmethod
    def flatten_state_dict(arch: Arch, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
        parts = []
        for name, (o, i) in arch.param_shapes():
            w, b = sd[name + ".weight"], sd[name + ".bias"]
            assert tuple(w.shape) == (o, i), (name, tuple(w.shape), (o, i))
            parts += [w.reshape(-1), b.reshape(-1)]
        return torch.cat(parts).contiguous()

    @staticmethod
    def init_flat(arch: Arch, seed: int = 42) -> torch.Tensor:
        """torch.nn.Linear default init (kaiming_uniform a=sqrt(5) == U(-1/sqrt(in), 1/sqrt(in)))."""
        g = torch.Generator().manual_seed(seed)
        parts = []
        for _, (o, i) in arch.param_shapes():
            bound = 1.0 / math.sqrt(i)
            parts.append((torch.rand(o * i, generator=g) * 2 - 1) * bound)
            parts.append((torch.rand(o, generator=g) * 2 - 1) * bound)
        return torch.cat(parts).contiguous()

    # -- forward ----------------------------------------------------------------------------
    def _lin(self, name: str, x: torch.Tensor) -> torch.Tensor:
        w, b = self.p[name + ".weight"], self.p[name + ".bias"]
        if self.emu and self.act_format != "bf16":
            return _LinearSaved8.apply(x, w, b, self.act_format, self.dz_scale)
        if self.emu:
            w, x = _bf16r(w), _bf16r(x)
            return _RoundGradBf16.apply(x @ w.t() + b)
        return x @ w.t() + b

    def __call__(self, x_enc: torch.Tensor, d_enc: Optional[torch.Tensor]) -> torch.Tensor:
        """x_enc [P, in_xyz], d_enc [P, in_dir] (None without view directions) -> raw [P, 4] =
        (rgb_raw(3), sigma_raw(1))."""
        a = self.arch
        flex = a.lineage == "flexible"
        tname = (lambda i: "layer1" if i == 0 else f"layers_xyz.{i - 1}") if flex else (lambda i: f"pts_linears.{i}")
        h = x_enc
        for i in range(a.depth):
            if a.skip >= 0 and i == a.skip + 1:
                h = torch.cat([x_enc, h], -1)
            h = self._lin(tname(i), h)
            if i > 0 or a.relu_first:
                h = torch.relu(h)
        if not a.use_viewdirs:
            return self._lin("fc_out" if flex else "output_linear", h)
        sigma = self._lin("fc_alpha" if flex else "alpha_linear", h)
        feat = self._lin("fc_feat" if flex else "feature_linear", h)
        if a.relu_feat:
            feat = torch.relu(feat)
        g = torch.relu(self._lin("layers_dir.0" if flex else "views_linears.0", torch.cat([feat, d_enc], -1)))
        rgb = self._lin("fc_rgb" if flex else "rgb_linear", g)
        return torch.cat([rgb, sigma], -1)


# --------------------------------------------------------------------------------------------
# compositing and hierarchical sampling
# --------------------------------------------------------------------------------------------
def raw2outputs(raw: torch.Tensor, z_vals: torch.Tensor, rays_d: torch.Tensor,
                noise: Optional[torch.Tensor] = None, white_background: bool = False):
    """Alpha compositing.  raw [R,n,4], z_vals [R,n], rays_d [R,3]; ``noise`` [R,n] already scaled.

    delta_i = z_{i+1}-z_i (last 1e10), times |rays_d|; sigma = relu(raw_sigma + noise);
    alpha = 1-exp(-sigma delta); T_i = prod_{j<i}(1-alpha_j+1e-10); w = alpha T.
    Returns rgb [R,3], disp [R], acc [R], weights [R,n], depth [R].
    """
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    s = raw[..., 3] if noise is None else raw[..., 3] + noise
    alpha = 1.0 - torch.exp(-torch.relu(s) * dists)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1]), 1.0 - alpha + 1e-10], -1), -1)[..., :-1]
    weights = alpha * trans
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    acc_map = torch.sum(weights, -1)
    disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)
    if white_background:
        rgb_map = rgb_map + (1.0 - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map


def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, n_samples: int, det: bool = True,
               u: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Inverse-CDF sampling.  bins [R,B], weights [R,B-1] -> samples [R,n_samples].

    ``u`` overrides the uniform draws (used to replay the hash stream of the HIP path).
    """
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        if det:
            u = torch.linspace(0.0, 1.0, n_samples, dtype=bins.dtype).expand(list(cdf.shape[:-1]) + [n_samples])
        else:
            u = torch.rand(list(cdf.shape[:-1]) + [n_samples], dtype=bins.dtype)
    u = u.contiguous()
    inds = torch.searchsorted(cdf.contiguous(), u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    bin_b, bin_a = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    return bin_b + t * (bin_a - bin_b)


def stratified_z(ray_batch: torch.Tensor, n_coarse: int, lindisp: bool = False,
                 u: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Coarse depths.  ``u`` [R,n_coarse] in [0,1) switches stratified jitter on."""
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]
    t = torch.linspace(0.0, 1.0, n_coarse, dtype=ray_batch.dtype)
    if not lindisp:
        z = near * (1.0 - t) + far * t
    else:
        z = 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
    if u is not None:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * u
    return z


@dataclass
class RenderOpts:
    """The ``nerf.train`` / ``nerf.validation`` knobs (config.yml:46-63)."""
    num_coarse: int = 64
    num_fine: int = 128
    perturb: bool = False
    lindisp: bool = False
    radiance_field_noise_std: float = 0.0
    white_background: bool = False
    seed: int = 0


def _run_net(net: NeRFNet, ray_batch: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    a = net.arch
    o, d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    pts = o[:, None, :] + d[:, None, :] * z[..., None]
    R, n = z.shape
    x_enc = positional_encoding(pts.reshape(-1, 3), a.n_freq_xyz, a.include_input_xyz, a.log_sampling_xyz)
    d_enc = None
    if a.use_viewdirs:
        vd = ray_batch[:, 8:11]
        d_enc = positional_encoding(vd[:, None, :].expand(R, n, 3).reshape(-1, 3), a.n_freq_dir, a.include_input_dir,
                                    a.log_sampling_dir)
    return net(x_enc, d_enc).reshape(R, n, 4)


def render_rays(ray_batch: torch.Tensor, coarse: NeRFNet, fine: Optional[NeRFNet], opts: RenderOpts,
                ray_offset: int = 0) -> Dict[str, torch.Tensor]:
    """Volume-render a batch of rays.  ray_batch [R,11] ([R,8] for networks without view directions)
    -> dict with fine and coarse maps.

    Random draws (only when ``perturb`` / noise are on) come from the shared hash streams,
    indexed ``ray*n + sample``.  ``ray_offset``: index of the first row inside a larger batch, so
    that a large batch can be evaluated piece by piece with the draws of the whole batch.
    """
    R = ray_batch.shape[0]
    dt = ray_batch.dtype
    Nc, Nf = opts.num_coarse, opts.num_fine
    rays_d = ray_batch[:, 3:6]
    u_strat = None
    if opts.perturb:
        idx = np.arange(R * Nc, dtype=np.uint64) + np.uint64(ray_offset * Nc)
        u_strat = torch.from_numpy(rand_uniform(opts.seed, STREAM_STRATIFIED, idx)).to(dt).view(R, Nc)
    z = stratified_z(ray_batch, Nc, opts.lindisp, u_strat)

    def noise_for(stream, n):
        if opts.radiance_field_noise_std <= 0.0:
            return None
        idx = np.arange(R * n, dtype=np.uint64) + np.uint64(ray_offset * n)
        return torch.from_numpy(rand_normal(opts.seed, stream, idx)).to(dt).view(R, n) * opts.radiance_field_noise_std

    raw_c = _run_net(coarse, ray_batch, z)
    rgb_c, disp_c, acc_c, w_c, depth_c = raw2outputs(raw_c, z, rays_d, noise_for(STREAM_NOISE_COARSE, Nc),
                                                     opts.white_background)
    out = {"rgb_coarse": rgb_c, "disp_coarse": disp_c, "acc_coarse": acc_c, "depth_coarse": depth_c,
           "z_coarse": z, "weights_coarse": w_c, "raw_coarse": raw_c}
    if Nf > 0 and fine is not None:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        u_pdf = None
        if opts.perturb:
            idx = np.arange(R * Nf, dtype=np.uint64) + np.uint64(ray_offset * Nf)
            u_pdf = torch.from_numpy(rand_uniform(opts.seed, STREAM_PDF, idx)).to(dt).view(R, Nf)
        z_f = sample_pdf(mids, w_c[..., 1:-1].detach(), Nf, det=not opts.perturb, u=u_pdf).detach()
        z_all, _ = torch.sort(torch.cat([z, z_f], -1), -1)
        raw_f = _run_net(fine, ray_batch, z_all)
        rgb_f, disp_f, acc_f, w_f, depth_f = raw2outputs(raw_f, z_all, rays_d,
                                                         noise_for(STREAM_NOISE_FINE, Nc + Nf),
                                                         opts.white_background)
        out.update({"rgb_fine": rgb_f, "disp_fine": disp_f, "acc_fine": acc_f, "depth_fine": depth_f,
                    "z_fine_samples": z_f, "z_all": z_all, "weights_fine": w_f, "raw_fine": raw_f})
    return out


def nerf_loss(out: Dict[str, torch.Tensor], target: torch.Tensor) -> torch.Tensor:
    """mse(rgb_coarse, target) + mse(rgb_fine, target); psnr = -10 log10(loss) is logged from it."""
    loss = torch.mean((out["rgb_coarse"] - target) ** 2)
    if "rgb_fine" in out:
        loss = loss + torch.mean((out["rgb_fine"] - target) ** 2)
    return loss


def adam_step(p, g, m, v, step: int, lr: float, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam's update written out (the stored optimiser state in the reference
    checkpoints is this optimiser's: betas (0.9, 0.999), eps 1e-8, no weight decay)."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def lr_at(iteration: int, lr0: float, decay_factor: float = 0.1, lr_decay: int = 250) -> float:
    """lr_i = lr0 * factor^(i / (lr_decay*1000)); reproduces the lr stored in both checkpoints
    (config.yml:64-69 with iter 110000 -> 0.0018153902738505068)."""
    return lr0 * (decay_factor ** (iteration / (lr_decay * 1000)))


def spherical_pose(theta_deg: float, phi_deg: float, radius: float) -> torch.Tensor:
    """Camera-to-world on a sphere looking at the origin (Blender scene convention, z up)."""
    def trans_t(t):
        return torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, t], [0, 0, 0, 1]], dtype=torch.float64)

    def rot_phi(p):
        return torch.tensor([[1, 0, 0, 0], [0, math.cos(p), -math.sin(p), 0],
                             [0, math.sin(p), math.cos(p), 0], [0, 0, 0, 1]], dtype=torch.float64)

    def rot_theta(t):
        return torch.tensor([[math.cos(t), 0, -math.sin(t), 0], [0, 1, 0, 0],
                             [math.sin(t), 0, math.cos(t), 0], [0, 0, 0, 1]], dtype=torch.float64)
    c2w = trans_t(radius)
    c2w = rot_phi(phi_deg / 180.0 * math.pi) @ c2w
    c2w = rot_theta(theta_deg / 180.0 * math.pi) @ c2w
    flip = torch.tensor([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=torch.float64)
    return (flip @ c2w).to(torch.float32)


def sample_training_rays(H: int, W: int, focal: float, c2w: torch.Tensor, image_u8: np.ndarray, white: bool,
                         near: float, far: float, seed: int, counter: int, R: int):
    """Ray selection of one training iteration on one view: pixel of ray r = floor(u32 * H*W / 2^32)
    with u32 = element counter + r of RNG stream STREAM_PIXELS (drawn with replacement, like the
    train loop's random pixel choice); returns (pixel indices, ray batch [R, 11], target [R, 3])."""
    draws = rand_u32(seed, STREAM_PIXELS, np.arange(R, dtype=np.uint64) + np.uint64(counter)).astype(np.uint64)
    pix = ((draws * np.uint64(H * W)) >> np.uint64(32)).astype(np.int64)
    ro, rd = get_rays(H, W, focal, c2w)
    rb = make_ray_batch(ro, rd, near, far)[torch.from_numpy(pix)]
    px = torch.from_numpy(np.asarray(image_u8).reshape(H * W, -1)[pix].astype(np.float32)) / 255.0
    if px.shape[1] == 4 and white:
        target = px[:, :3] * px[:, 3:] + (1.0 - px[:, 3:])
    else:
        target = px[:, :3]
    return pix, rb, target.contiguous()
