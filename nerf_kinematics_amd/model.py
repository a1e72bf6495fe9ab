"""Network description and parameter container for the HIP render path.

Two nerf-pytorch lineages are supported (``lineage``):

``run_nerf``   the 8x256 network of ``run_nerf.py`` (``pts_linears``, ``views_linears``,
               ``feature_linear``, ``alpha_linear``, ``rgb_linear``);
``flexible``   the YAML-configured ``FlexibleNeRFModel`` the reference trained
               (``models.coarse.type`` in
               /root/reference/trained_models/nerf-pytorch/lego-lowres3/config.yml:29) with the
               key names of its checkpoints (``layer1``, ``layers_xyz.N``, ``layers_dir.0``,
               ``fc_alpha``, ``fc_rgb``, ``fc_feat``).

``use_viewdirs=False`` (``models.*.use_viewdirs`` of the same file, lines 30 and 42) drops the direction
branch of either lineage: the trunk is followed by ONE linear layer with four rows
(rgb_raw, sigma_raw), ``output_linear`` / ``fc_out`` in the public nerf-pytorch sources.

Parameters live in ONE flat fp32 tensor in state-dict order; the MFMA kernels read a bf16
fragment image of it (``NeRFModel.pack``), refreshed after every optimiser step.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch

from . import lib as _lib


@dataclass(frozen=True)
class Arch:
    depth: int = 8
    width: int = 256
    skip: int = 4
    dir_width: int = 128
    n_freq_xyz: int = 10
    n_freq_dir: int = 4
    relu_first: bool = True
    relu_feat: bool = False
    lineage: str = "run_nerf"
    use_viewdirs: bool = True
    # include_input_xyz / _dir, log_sampling_xyz / _dir of the models block (config.yml:21-24)
    include_input_xyz: bool = True
    include_input_dir: bool = True
    log_sampling_xyz: bool = True
    log_sampling_dir: bool = True

    @property
    def in_xyz(self) -> int:
        return (3 if self.include_input_xyz else 0) + 6 * self.n_freq_xyz

    @property
    def in_dir(self) -> int:
        return (3 if self.include_input_dir else 0) + 6 * self.n_freq_dir

    @property
    def enc_flags(self) -> int:
        return ((0 if self.include_input_xyz else _lib.NK_ENC_NO_INPUT_XYZ) | (0 if self.include_input_dir else _lib.NK_ENC_NO_INPUT_DIR)
                | (0 if self.log_sampling_xyz else _lib.NK_ENC_LINEAR_XYZ) | (0 if self.log_sampling_dir else _lib.NK_ENC_LINEAR_DIR))

    def to_c(self) -> _lib.NkArch:
        return _lib.NkArch(self.depth, self.width, self.skip, self.dir_width, self.n_freq_xyz, self.n_freq_dir,
                           int(self.relu_first), int(self.relu_feat),
                           _lib.NK_LINEAGE_FLEXIBLE if self.lineage == "flexible" else _lib.NK_LINEAGE_RUN_NERF,
                           int(not self.use_viewdirs), self.enc_flags)

    def param_shapes(self) -> List[Tuple[str, Tuple[int, int]]]:
        """[(module name, (out, in))] in state-dict order."""
        W, D = self.width, self.depth
        trunk = []
        for i in range(D):
            k = self.in_xyz if i == 0 else (W + self.in_xyz if (self.skip >= 0 and i == self.skip + 1) else W)
            trunk.append((W, k))
        dirs = (self.dir_width, W + self.in_dir)
        if self.lineage == "flexible":
            names = [("layer1", trunk[0])] + [(f"layers_xyz.{i - 1}", trunk[i]) for i in range(1, D)]
            if not self.use_viewdirs:
                return names + [("fc_out", (4, W))]
            names += [("layers_dir.0", dirs), ("fc_alpha", (1, W)), ("fc_rgb", (3, self.dir_width)), ("fc_feat", (W, W))]
        else:
            names = [(f"pts_linears.{i}", trunk[i]) for i in range(D)]
            if not self.use_viewdirs:
                return names + [("output_linear", (4, W))]
            names += [("views_linears.0", dirs), ("feature_linear", (W, W)), ("alpha_linear", (1, W)),
                      ("rgb_linear", (3, self.dir_width))]
        return names

    def num_params(self) -> int:
        return sum(o * i + o for _, (o, i) in self.param_shapes())

    @staticmethod
    def from_config(model_cfg: dict, state_dict: Dict[str, torch.Tensor] | None = None) -> "Arch":
        """``models.coarse`` / ``models.fine`` block of a reference config.yml -> Arch.

        The vendored checkpoints hold fewer trunk layers than ``num_layers`` says (4, no skip), so
        when a state dict is given its keys decide the depth.
        """
        typ = model_cfg.get("type", "FlexibleNeRFModel")
        W = int(model_cfg.get("hidden_size", 256))
        lx = int(model_cfg.get("num_encoding_fn_xyz", 10))
        ld = int(model_cfg.get("num_encoding_fn_dir", 4))
        vd = bool(model_cfg.get("use_viewdirs", True))
        enc = dict(include_input_xyz=bool(model_cfg.get("include_input_xyz", True)),
                   include_input_dir=bool(model_cfg.get("include_input_dir", True)),
                   log_sampling_xyz=bool(model_cfg.get("log_sampling_xyz", True)),
                   log_sampling_dir=bool(model_cfg.get("log_sampling_dir", True)))
        if state_dict is not None:      # a stored network says itself whether it has a direction branch
            vd = not any(k in state_dict for k in ("fc_out.weight", "output_linear.weight"))
        if typ == "FlexibleNeRFModel":
            # The reference's training script builds this class with its default depth whatever the
            # file says: both vendored checkpoints hold layer1 + three layers_xyz under
            # ``num_layers: 8, skip_connect_every: 3``.  ``honour_num_layers: true`` opts into the
            # depth the file names (no skip connection in this lineage's kernels).
            depth = 4
            if state_dict is not None:
                depth = 1 + sum(1 for k in state_dict if k.startswith("layers_xyz.") and k.endswith(".weight"))
            elif model_cfg.get("honour_num_layers"):
                depth = int(model_cfg.get("num_layers", 4))
            elif int(model_cfg.get("num_layers", 4)) != 4:
                import warnings
                warnings.warn(f"FlexibleNeRFModel: num_layers: {model_cfg.get('num_layers')} / skip_connect_every are not "
                              "honoured (the reference trains 4 trunk layers whatever they say); building 4 layers. "
                              "Set honour_num_layers: true in the model block to build the depth named.", stacklevel=2)
            return Arch(depth=depth, width=W, skip=-1, dir_width=W // 2, n_freq_xyz=lx, n_freq_dir=ld,
                        relu_first=False, relu_feat=True, lineage="flexible", use_viewdirs=vd, **enc)
        if typ in ("NeRF", "run_nerf"):
            D = int(model_cfg.get("num_layers", 8))
            return Arch(depth=D, width=W, skip=int(model_cfg.get("skip", 4 if D > 5 else -1)), dir_width=W // 2,
                        n_freq_xyz=lx, n_freq_dir=ld, relu_first=True, relu_feat=False, lineage="run_nerf",
                        use_viewdirs=vd, **enc)
        raise ValueError(f"unknown model type {typ!r}")


ARCH_RUN_NERF = Arch()
ARCH_LEGO_CKPT = Arch(4, 128, -1, 64, 10, 4, False, True, "flexible")
ARCH_WHEEL_CKPT = Arch(4, 128, -1, 64, 6, 4, False, True, "flexible")


class NeRFModel:
    """Flat fp32 parameters (+ bf16 fragment image) of one network on one GPU."""

    def __init__(self, arch: Arch, flat: torch.Tensor):
        L = _lib.load()
        self.arch = arch
        self._c = arch.to_c()
        n = _lib.check(L.nk_num_params(self._c))
        if flat.dtype != torch.float32 or flat.dim() != 1 or flat.numel() != n:
            raise ValueError(f"flat parameters must be float32[{n}], got {flat.dtype}{tuple(flat.shape)}")
        if not flat.is_cuda:
            raise ValueError("NeRFModel parameters must live on the GPU (there is no CPU render path)")
        self.flat = flat.contiguous()
        self.packed = torch.empty(_lib.check(L.nk_packed_bytes(self._c)), dtype=torch.uint8, device=flat.device)
        self._packed_version = None
        self.pack()

    # -- construction ---------------------------------------------------------------------
    @staticmethod
    def init(arch: Arch, device, seed: int = 42) -> "NeRFModel":
        """torch.nn.Linear default initialisation, U(-1/sqrt(in), 1/sqrt(in)) for weight and bias."""
        g = torch.Generator().manual_seed(seed)
        parts = []
        for _, (o, i) in arch.param_shapes():
            b = 1.0 / math.sqrt(i)
            parts.append((torch.rand(o * i, generator=g) * 2 - 1) * b)
            parts.append((torch.rand(o, generator=g) * 2 - 1) * b)
        return NeRFModel(arch, torch.cat(parts).to(device))

    @staticmethod
    def from_state_dict(arch: Arch, sd: Dict[str, torch.Tensor], device) -> "NeRFModel":
        parts = []
        for name, (o, i) in arch.param_shapes():
            w, b = sd[name + ".weight"], sd[name + ".bias"]
            if tuple(w.shape) != (o, i) or tuple(b.shape) != (o,):
                raise ValueError(f"{name}: checkpoint shape {tuple(w.shape)} != {(o, i)}")
            parts += [w.reshape(-1).float(), b.reshape(-1).float()]
        return NeRFModel(arch, torch.cat(parts).to(device))

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Named views in the reference's key order (CPU copies)."""
        out, off = {}, 0
        flat = self.flat.detach().cpu()
        for name, (o, i) in self.arch.param_shapes():
            out[name + ".weight"] = flat[off:off + o * i].view(o, i).clone(); off += o * i
            out[name + ".bias"] = flat[off:off + o].clone(); off += o
        return out

    # -- bf16 fragment image -----------------------------------------------------------------
    def pack(self) -> None:
        L = _lib.load()
        stream = torch.cuda.current_stream(self.flat.device).cuda_stream
        with torch.cuda.device(self.flat.device):
            _lib.check(L.nk_pack_weights(self._c, self.flat.data_ptr(), self.packed.data_ptr(), stream))
        self._packed_version = self.flat._version

    def ensure_packed(self) -> None:
        if self._packed_version != self.flat._version:
            self.pack()
