"""Experiment configuration in the reference's YAML schema.

The schema is the one of the experiment files the reference ships next to its checkpoints
(/root/reference/trained_models/nerf-pytorch/lego-lowres3/config.yml:1-69 and
wheel_robot6/config.yml): blocks ``dataset`` (:1-9), ``experiment`` (:10-17), ``models`` (:18-42),
``nerf`` (:43-63), ``optimizer`` (:64-66), ``scheduler`` (:67-69).  A file written for the
reference loads here unchanged; missing keys take the values of the lego file.
"""
from __future__ import annotations

import copy
from typing import Any, Dict, Optional

import yaml

_MODEL_DEFAULT = {
    "type": "FlexibleNeRFModel", "hidden_size": 128, "num_layers": 8, "skip_connect_every": 3,
    "num_encoding_fn_xyz": 10, "num_encoding_fn_dir": 4, "include_input_xyz": True, "include_input_dir": True,
    "log_sampling_xyz": True, "log_sampling_dir": True, "use_viewdirs": True,
}
_RENDER_DEFAULT = {
    "chunksize": 131072, "lindisp": False, "num_coarse": 64, "num_fine": 64, "perturb": False,
    "radiance_field_noise_std": 0.0, "white_background": False,
}
# Two keys the reference's files do not have, both experiments that are off unless a file names them:
#   nerf.train.act_format            "bf16" | "fp8" | "fp8bf8": format of the tensors saved for backward
#   nerf.validation.skip_empty_eps   > 0: renders skip the fine network where the coarse pass saw nothing
ACT_FORMATS = ("bf16", "fp8", "fp8bf8")
DEFAULTS: Dict[str, Any] = {
    "dataset": {"type": "blender", "basedir": "cache/nerf_synthetic/lego", "cachedir": None, "near": 2, "far": 6,
                "half_res": True, "no_ndc": True, "testskip": 1, "downsample_factor": 8, "llffhold": 8},
    "experiment": {"id": "experiment", "logdir": "logs", "print_every": 100, "randomseed": 42, "save_every": 5000,
                   "train_iters": 200000, "validate_every": 100},
    "models": {"coarse": dict(_MODEL_DEFAULT), "fine": dict(_MODEL_DEFAULT)},
    "nerf": {"encode_position_fn": "positional_encoding", "encode_direction_fn": "positional_encoding",
             "use_viewdirs": True,
             "train": dict(_RENDER_DEFAULT, num_random_rays=1024, perturb=True, radiance_field_noise_std=0.2),
             "validation": dict(_RENDER_DEFAULT)},
    "optimizer": {"type": "Adam", "lr": 0.005},
    "scheduler": {"lr_decay": 250, "lr_decay_factor": 0.1},
}


class ConfigError(ValueError):
    pass


def _merge(base: dict, over: dict, path: str = "") -> dict:
    out = copy.deepcopy(base)
    for k, v in (over or {}).items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = _merge(out[k], v, f"{path}{k}.")
        else:
            out[k] = copy.deepcopy(v)
    return out


def validate(cfg: dict) -> dict:
    """Reject what the HIP render path cannot run, naming the key (plain ValueError subclass,
    raised before anything touches the GPU)."""
    nerf = cfg["nerf"]
    for fn in ("encode_position_fn", "encode_direction_fn"):
        if nerf.get(fn, "positional_encoding") != "positional_encoding":
            raise ConfigError(f"nerf.{fn}: only 'positional_encoding' is supported, got {nerf[fn]!r}")
    vd = bool(nerf.get("use_viewdirs", True))
    for which in ("coarse", "fine"):
        m = cfg["models"].get(which)
        if m is None:
            if which == "coarse":
                raise ConfigError("models.coarse is required")
            continue
        # the ray rows carry view directions exactly when the networks take them (in the reference a
        # mismatch is a shape error at the first layer of the direction branch)
        if bool(m.get("use_viewdirs", True)) != vd:
            raise ConfigError(f"models.{which}.use_viewdirs: {m.get('use_viewdirs', True)} contradicts nerf.use_viewdirs: {vd}")
        if m.get("type") not in ("FlexibleNeRFModel", "NeRF", "run_nerf"):
            raise ConfigError(f"models.{which}.type: unknown model type {m.get('type')!r}")
        if int(m["hidden_size"]) not in (128, 256):
            raise ConfigError(f"models.{which}.hidden_size: 128 or 256, got {m['hidden_size']}")
        if not 1 <= int(m["num_encoding_fn_xyz"]) <= 10:
            raise ConfigError(f"models.{which}.num_encoding_fn_xyz: 1..10, got {m['num_encoding_fn_xyz']}")
        if not 1 <= int(m["num_encoding_fn_dir"]) <= 4:
            raise ConfigError(f"models.{which}.num_encoding_fn_dir: 1..4, got {m['num_encoding_fn_dir']}")
    if nerf["train"].get("act_format", "bf16") not in ACT_FORMATS:
        raise ConfigError(f"nerf.train.act_format: one of {ACT_FORMATS}, got {nerf['train']['act_format']!r}")
    if not float(nerf["validation"].get("skip_empty_eps", 0.0)) >= 0.0:
        raise ConfigError(f"nerf.validation.skip_empty_eps: >= 0, got {nerf['validation']['skip_empty_eps']!r}")
    for block in ("train", "validation"):
        r = nerf[block]
        if int(r["num_coarse"]) < 2:
            raise ConfigError(f"nerf.{block}.num_coarse must be >= 2")
        if int(r["num_fine"]) < 0:
            raise ConfigError(f"nerf.{block}.num_fine must be >= 0")
    if cfg["optimizer"].get("type", "Adam") != "Adam":
        raise ConfigError(f"optimizer.type: only Adam is supported, got {cfg['optimizer']['type']!r}")
    if cfg["dataset"]["type"] not in ("blender", "robot", "transforms", "teacher", "forward_grid"):
        raise ConfigError(f"dataset.type: unknown dataset type {cfg['dataset']['type']!r}")
    return cfg


def load_config(path: str, overrides: Optional[dict] = None) -> dict:
    """Read a reference config.yml (safe loader) and fill in defaults."""
    with open(path, "r") as fh:
        raw = yaml.safe_load(fh) or {}
    if not isinstance(raw, dict):
        raise ConfigError(f"{path}: top level must be a mapping")
    cfg = _merge(DEFAULTS, raw)
    if "fine" not in (raw.get("models") or {}) and "models" in raw:
        cfg["models"]["fine"] = None          # a file that names only a coarse network has no fine one
    if overrides:
        cfg = _merge(cfg, overrides)
    return validate(cfg)


def dump_config(cfg: dict, path: str) -> None:
    """Write the configuration next to the checkpoints, as the reference's runs do."""
    with open(path, "w") as fh:
        yaml.safe_dump(cfg, fh, default_flow_style=False, sort_keys=True)


def render_kwargs(cfg: dict, block: str) -> dict:
    """``nerf.train`` / ``nerf.validation`` -> keyword arguments of ``render_rays``."""
    r = cfg["nerf"][block]
    fine = int(r["num_fine"]) if cfg["models"].get("fine") is not None else 0
    kw = dict(num_coarse=int(r["num_coarse"]), num_fine=fine, perturb=bool(r["perturb"]),
              lindisp=bool(r["lindisp"]), radiance_field_noise_std=float(r["radiance_field_noise_std"]),
              white_background=bool(r["white_background"]))
    if block == "validation" and float(r.get("skip_empty_eps", 0.0)) > 0.0:
        kw["skip_empty_eps"] = float(r["skip_empty_eps"])
    return kw
