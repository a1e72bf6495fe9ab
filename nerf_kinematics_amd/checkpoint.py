"""Reading and writing checkpoints in the reference's ``.ckpt`` layout.

Layout (as found in
/root/reference/trained_models/nerf-pytorch/lego-lowres3/checkpoint110000.ckpt, a ``torch.save``
dict): ``iter`` (int), ``model_coarse_state_dict`` / ``model_fine_state_dict`` (ordered
name -> tensor), ``optimizer_state_dict`` (torch.optim.Adam: ``state[i] = {step (float32
scalar), exp_avg, exp_avg_sq}`` for parameter i, coarse parameters first then fine, each in
state-dict order; one ``param_groups`` entry), ``loss`` (0-d float32 tensor), ``psnr`` (float,
``-10 log10(loss)``).  Files written here load back into upstream code with ``torch.load`` +
``load_state_dict`` and files written by the reference load here.

Everything in this module works on CPU tensors; the trainer moves flat buffers to and from the
GPU.  Checkpoints are only ever read with ``weights_only=True``.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from .model import Arch


def split_flat(arch: Arch, flat: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
    """Flat fp32 parameter vector -> named tensors in the reference's key order."""
    flat = flat.detach().to("cpu", torch.float32)
    if flat.numel() != arch.num_params():
        raise ValueError(f"flat vector has {flat.numel()} values, the network {arch.num_params()}")
    out, off = OrderedDict(), 0
    for name, (o, i) in arch.param_shapes():
        out[name + ".weight"] = flat[off:off + o * i].view(o, i).clone(); off += o * i
        out[name + ".bias"] = flat[off:off + o].clone(); off += o
    return out


def join_state_dict(arch: Arch, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Named tensors -> flat fp32 vector; a strict load (missing, unexpected or mis-shaped keys raise)."""
    want = []
    for name, (o, i) in arch.param_shapes():
        want += [(name + ".weight", (o, i)), (name + ".bias", (o,))]
    missing = [k for k, _ in want if k not in sd]
    extra = [k for k in sd if k not in dict(want)]
    if missing or extra:
        raise ValueError(f"state dict does not match the network: missing {missing}, unexpected {extra}")
    parts = []
    for k, shape in want:
        if tuple(sd[k].shape) != shape:
            raise ValueError(f"{k}: checkpoint shape {tuple(sd[k].shape)} != {shape}")
        parts.append(sd[k].detach().reshape(-1).to("cpu", torch.float32))
    return torch.cat(parts)


def arch_for_state_dict(model_cfg: dict, sd: Dict[str, torch.Tensor]) -> Arch:
    """Network description from a ``models.*`` block, with depth / widths taken from the tensors
    (the vendored checkpoints hold 4 trunk layers although their config says ``num_layers: 8``)."""
    arch = Arch.from_config(model_cfg, sd)
    first = "layer1.weight" if arch.lineage == "flexible" else "pts_linears.0.weight"
    if first not in sd:
        raise ValueError(f"state dict has no {first}; keys: {list(sd)[:4]}...")
    W, in_xyz = sd[first].shape
    raw_x, raw_d = (3 if arch.include_input_xyz else 0), (3 if arch.include_input_dir else 0)
    if (in_xyz - raw_x) % 6:
        raise ValueError(f"{first}: input width {in_xyz} is not {raw_x} + 6 L")
    if arch.use_viewdirs:
        dirs = "layers_dir.0.weight" if arch.lineage == "flexible" else "views_linears.0.weight"
        WD, in_views = sd[dirs].shape
    else:       # no direction branch: its width and encoding are what the model block says (unused)
        WD, in_views = arch.dir_width, int(W) + arch.in_dir
    skip = arch.skip
    if arch.lineage == "run_nerf":
        skip = -1
        for i in range(arch.depth):
            if sd[f"pts_linears.{i}.weight"].shape[1] == W + in_xyz:
                skip = i - 1
    return Arch(arch.depth, int(W), skip, int(WD), (int(in_xyz) - raw_x) // 6, (int(in_views) - int(W) - raw_d) // 6,
                arch.relu_first, arch.relu_feat, arch.lineage, arch.use_viewdirs, arch.include_input_xyz,
                arch.include_input_dir, arch.log_sampling_xyz, arch.log_sampling_dir)


@dataclass
class TrainState:
    """Everything a checkpoint carries, as CPU tensors."""
    iteration: int
    coarse: torch.Tensor                    # flat parameters
    fine: Optional[torch.Tensor]
    exp_avg: List[torch.Tensor]             # [coarse, fine] flat first moments (may be empty)
    exp_avg_sq: List[torch.Tensor]
    opt_step: int                           # optimiser steps taken
    lr: float
    loss: float = float("nan")
    psnr: float = float("nan")


def load_checkpoint(path: str) -> dict:
    """The raw dict of a ``.ckpt`` (tensors on CPU).  Never unpickles arbitrary objects."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    for key in ("iter", "model_coarse_state_dict"):
        if key not in ck:
            raise ValueError(f"{path}: not a NeRF checkpoint (no {key!r})")
    return ck


def state_from_checkpoint(ck: dict, arch_coarse: Arch, arch_fine: Optional[Arch]) -> TrainState:
    coarse = join_state_dict(arch_coarse, ck["model_coarse_state_dict"])
    fine = None
    if arch_fine is not None and ck.get("model_fine_state_dict"):
        fine = join_state_dict(arch_fine, ck["model_fine_state_dict"])
    archs = [arch_coarse] + ([arch_fine] if fine is not None else [])
    m, v, step, lr = [], [], 0, float("nan")
    opt = ck.get("optimizer_state_dict")
    if opt and opt.get("state"):
        st, idx = opt["state"], 0
        for arch in archs:
            pm, pv = [], []
            for _, (o, i) in arch.param_shapes():
                for shape in ((o, i), (o,)):
                    e = st[idx]
                    if tuple(e["exp_avg"].shape) != shape:
                        raise ValueError(f"optimizer state {idx}: shape {tuple(e['exp_avg'].shape)} != {shape}")
                    pm.append(e["exp_avg"].reshape(-1).float()); pv.append(e["exp_avg_sq"].reshape(-1).float())
                    step = int(float(e["step"]))
                    idx += 1
            m.append(torch.cat(pm)); v.append(torch.cat(pv))
        if idx != len(st):
            raise ValueError(f"optimizer state holds {len(st)} parameters, the networks {idx}")
        lr = float(opt["param_groups"][0]["lr"])
    loss = float(ck["loss"]) if "loss" in ck else float("nan")
    return TrainState(int(ck["iter"]), coarse, fine, m, v, step, lr, loss, float(ck.get("psnr", float("nan"))))


def checkpoint_from_state(s: TrainState, arch_coarse: Arch, arch_fine: Optional[Arch],
                          betas=(0.9, 0.999), eps: float = 1e-8) -> dict:
    ck = {"iter": int(s.iteration), "model_coarse_state_dict": split_flat(arch_coarse, s.coarse),
          "model_fine_state_dict": split_flat(arch_fine, s.fine) if s.fine is not None else None}
    archs = [arch_coarse] + ([arch_fine] if s.fine is not None else [])
    state, idx = {}, 0
    for k, arch in enumerate(archs):
        off = 0
        for _, (o, i) in arch.param_shapes():
            for shape in ((o, i), (o,)):
                n = int(math.prod(shape))
                entry = {"step": torch.tensor(float(s.opt_step), dtype=torch.float32)}
                if s.exp_avg:
                    entry["exp_avg"] = s.exp_avg[k][off:off + n].view(*shape).clone()
                    entry["exp_avg_sq"] = s.exp_avg_sq[k][off:off + n].view(*shape).clone()
                state[idx] = entry
                off += n; idx += 1
    group = {"lr": float(s.lr), "betas": tuple(betas), "eps": float(eps), "weight_decay": 0, "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(idx))}
    ck["optimizer_state_dict"] = {"state": state if s.exp_avg else {}, "param_groups": [group]}
    ck["loss"] = torch.tensor(float(s.loss), dtype=torch.float32)
    ck["psnr"] = float(s.psnr)
    return ck


def save_checkpoint(path: str, s: TrainState, arch_coarse: Arch, arch_fine: Optional[Arch], **adam) -> None:
    torch.save(checkpoint_from_state(s, arch_coarse, arch_fine, **adam), path)


def checkpoint_name(iteration: int) -> str:
    """``checkpoint110000.ckpt`` / ``checkpoint05000.ckpt``: five digits, zero padded."""
    return f"checkpoint{int(iteration):05d}.ckpt"
