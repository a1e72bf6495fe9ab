"""Scenes (posed images) and the per-iteration ray sampler that feed ``render_rays``.

Dataset kinds follow ``dataset.type`` of the reference's experiment files:

``blender``     nerf_synthetic layout: ``transforms_{train,val,test}.json`` with ``camera_angle_x``
                and ``frames[].file_path`` / ``transform_matrix``
                (/root/reference/trained_models/nerf-pytorch/lego-lowres3/config.yml:1-9:
                ``basedir``, ``half_res``, ``testskip``, ``near``, ``far``, ``no_ndc``).
``robot``       the reference's own captures (wheel_robot6/config.yml:1-8: ``downsample_factor``,
                ``llffhold``, forward-facing NDC with near 0 / far 1): ``transforms.json`` as
                written by the pose converter (``fl_x fl_y cx cy w h``,
                /root/reference/parser_instant_ngp.py:128-145), every ``llffhold``-th frame held
                out (the LLFF rule the key name comes from; the reference's loader for this type
                is not vendored).  Images are down-sampled by plain striding, ``img[::f, ::f]``:
                the reference's own 160 x 90 validation target
                (/root/reference/results/wheel/nerf-pytorch/imgs/reals/TestNERF_0_original.png)
                equals ``TestNERF 0.jpg[::8, ::8]`` to 44.5 dB (JPEG decoder differences), against
                20 dB for any filtered resize.
``forward_grid`` synthetic forward-facing capture (the 1280 x 720 NDC configuration; the fox
                photographs ship without poses): cameras on a square grid in the z = 0 plane
                looking down -z at the lego teacher placed at z = -4, NDC rays, near 0 / far 1.
``transforms``  any single transforms.json of either flavour.
``teacher``     synthetic: views of a trained checkpoint rendered by the HIP path itself (the
                nerf_synthetic photographs are not redistributable with the repository).

Images stay 8-bit in HBM ([N, H, W, C] uint8); a training iteration is one kernel
(``nk_sample_training_rays``) that draws the pixels, builds their rays and converts their colours,
so nothing crosses PCIe per step.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as _lib

_M32 = 0xFFFFFFFF


def _lowbias32(x: int) -> int:
    x &= _M32
    x ^= x >> 16; x = (x * 0x7FEB352D) & _M32
    x ^= x >> 15; x = (x * 0x846CA68B) & _M32
    x ^= x >> 16
    return x


def rand_u32(seed: int, stream: int, idx: int) -> int:
    """Host copy of the library's counter-based generator (same bits as the device function)."""
    key = _lowbias32((seed ^ ((stream * 0x85EBCA6B) & _M32)) & _M32)
    lo, hi = idx & _M32, (idx >> 32) & _M32
    x = (lo * 0x9E3779B9 + key) & _M32
    return _lowbias32(x ^ ((hi * 0xC2B2AE35) & _M32))


def spherical_pose(theta_deg: float, phi_deg: float, radius: float) -> np.ndarray:
    """4x4 camera-to-world on a sphere around the origin, looking at it; Blender scene axes (z up),
    OpenGL camera axes (x right, y up, looking down -z)."""
    th, ph = math.radians(theta_deg), math.radians(phi_deg)
    t = np.eye(4); t[2, 3] = radius
    rp = np.array([[1, 0, 0, 0], [0, math.cos(ph), -math.sin(ph), 0], [0, math.sin(ph), math.cos(ph), 0], [0, 0, 0, 1.0]])
    rt = np.array([[math.cos(th), 0, -math.sin(th), 0], [0, 1, 0, 0], [math.sin(th), 0, math.cos(th), 0], [0, 0, 0, 1.0]])
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]])
    return (flip @ rt @ rp @ t).astype(np.float32)


@dataclass
class Scene:
    images: Optional[torch.Tensor]        # uint8 [N, H, W, C] (C = 3 | 4); None for pose-only scenes
    poses: np.ndarray                     # float32 [N, 4, 4] camera-to-world
    H: int
    W: int
    fx: float
    fy: float
    cx: float
    cy: float
    near: float
    far: float
    ndc: bool = False
    white_background: bool = False
    i_train: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    i_val: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    i_test: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    names: List[str] = field(default_factory=list)

    def camera(self, k: int, pose: Optional[np.ndarray] = None) -> _lib.NkCamera:
        cam = _lib.NkCamera()
        cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy = self.H, self.W, self.fx, self.fy, self.cx, self.cy
        m = np.asarray(self.poses[k] if pose is None else pose, dtype=np.float32)
        for j, v in enumerate(m[:3, :4].reshape(-1).tolist()):
            cam.c2w[j] = v
        cam.near, cam.far, cam.ndc, cam.ndc_near = float(self.near), float(self.far), int(self.ndc), 1.0
        return cam

    def rays(self, k: int, device=None, pose: Optional[np.ndarray] = None) -> torch.Tensor:
        """Whole-image ray batch [H*W, 11] of view k (or of an explicit pose)."""
        from .render import _stream
        dev = torch.device(device) if device is not None else self.images.device
        out = torch.empty(self.H * self.W, 11, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().nk_get_rays(self.camera(k, pose), None, 0, self.H * self.W, out.data_ptr(),
                                               _stream(dev)))
        return out

    def target(self, k: int) -> torch.Tensor:
        """Colours of view k as float [H*W, 3] (evaluation side; training uses the fused sampler)."""
        img = self.images[k].reshape(-1, self.images.shape[-1]).float() / 255.0
        if img.shape[-1] == 4 and self.white_background:
            return img[:, :3] * img[:, 3:] + (1.0 - img[:, 3:])
        return img[:, :3].contiguous()


# ---------------------------------------------------------------------------------------------
# file loaders (host side: PIL + json)
# ---------------------------------------------------------------------------------------------
def _read_image(path: str, size: Optional[Tuple[int, int]] = None, stride: int = 0) -> np.ndarray:
    """8-bit image, optionally reduced: ``stride`` f > 1 keeps every f-th row and column (the
    reference's robot loader), otherwise ``size`` (W, H) by area averaging."""
    from PIL import Image
    with Image.open(path) as im:
        im = im.convert("RGBA" if im.mode in ("RGBA", "LA", "P") else "RGB")
        if stride > 1:
            return np.ascontiguousarray(np.asarray(im, dtype=np.uint8)[::stride, ::stride])
        if size is not None and im.size != size:
            im = im.resize(size, Image.BOX)          # area averaging (exact for integer factors)
        return np.asarray(im, dtype=np.uint8)


def _stack(imgs: Sequence[np.ndarray], device) -> torch.Tensor:
    shapes = {a.shape for a in imgs}
    if len(shapes) != 1:
        raise ValueError(f"images of one scene must share a size and channel count, got {sorted(shapes)}")
    return torch.from_numpy(np.stack(imgs)).to(device)


def _resolve(base: str, file_path: str, exts=(".png", ".jpg", ".jpeg", ".JPG")) -> str:
    p = file_path if os.path.isabs(file_path) else os.path.join(base, file_path)
    if os.path.isfile(p):
        return p
    for e in exts:
        if os.path.isfile(p + e):
            return p + e
    # a converter run from another directory stores paths relative to ITS working directory
    q = os.path.join(base, os.path.basename(os.path.dirname(file_path)), os.path.basename(file_path))
    if os.path.isfile(q):
        return q
    raise FileNotFoundError(f"image {file_path!r} not found under {base}")


def load_blender(basedir: str, half_res: bool = False, testskip: int = 1, near: float = 2.0, far: float = 6.0,
                 white_background: bool = False, device="cuda", splits=("train", "val", "test")) -> Scene:
    imgs, poses, names, counts, cax = [], [], [], [], None
    for s in splits:
        path = os.path.join(basedir, f"transforms_{s}.json")
        if not os.path.isfile(path):
            if s == "train":
                raise FileNotFoundError(path)
            counts.append(0)
            continue
        with open(path) as fh:
            meta = json.load(fh)
        cax = float(meta["camera_angle_x"]) if cax is None else cax
        skip = 1 if (s == "train" or testskip == 0) else testskip
        frames = meta["frames"][::skip]
        for fr in frames:
            p = _resolve(basedir, fr["file_path"])
            imgs.append(p); names.append(p)
            poses.append(np.array(fr["transform_matrix"], dtype=np.float32))
        counts.append(len(frames))
    from PIL import Image
    with Image.open(imgs[0]) as im:
        W, H = im.size
    if half_res:
        W, H = W // 2, H // 2
    data = [_read_image(p, (W, H)) for p in imgs]
    focal = 0.5 * W / math.tan(0.5 * cax)
    edges = np.cumsum([0] + counts)
    idx = [np.arange(edges[i], edges[i + 1]) for i in range(len(splits))]
    while len(idx) < 3:
        idx.append(np.zeros(0, np.int64))
    return Scene(_stack(data, device), np.stack(poses), H, W, focal, focal, W * 0.5, H * 0.5, near, far, False,
                 white_background, idx[0], idx[1], idx[2], names)


def read_transforms(path: str) -> Dict:
    """Intrinsics and frames of a transforms.json of either flavour (``camera_angle_x`` only, or
    ``fl_x fl_y cx cy w h``)."""
    with open(path) as fh:
        meta = json.load(fh)
    out = {"frames": meta.get("frames", []), "w": meta.get("w"), "h": meta.get("h"),
           "fl_x": meta.get("fl_x"), "fl_y": meta.get("fl_y"), "cx": meta.get("cx"), "cy": meta.get("cy"),
           "camera_angle_x": meta.get("camera_angle_x"), "camera_angle_y": meta.get("camera_angle_y"),
           "aabb_scale": meta.get("aabb_scale")}
    return out


def load_transforms(path: str, downsample: int = 1, near: float = 2.0, far: float = 6.0, ndc: bool = False,
                    holdout: int = 0, white_background: bool = False, device="cuda",
                    val_path: Optional[str] = None, load_images: bool = True, strided: bool = False) -> Scene:
    """One transforms.json (+ optionally a ``_val`` file) -> Scene.

    ``holdout`` > 0 holds out frames 0, holdout, 2*holdout, ... as the test/validation split (the
    LLFF rule); otherwise the frames of ``val_path`` are the validation split."""
    base = os.path.dirname(os.path.abspath(path))
    meta = read_transforms(path)
    frames = [f for f in meta["frames"] if "transform_matrix" in f]
    n_main = len(frames)
    if val_path and os.path.isfile(val_path):
        frames = frames + [f for f in read_transforms(val_path)["frames"] if "transform_matrix" in f]
    if not frames:
        raise ValueError(f"{path}: no frames with a transform_matrix")
    paths = [_resolve(base, f["file_path"]) for f in frames] if load_images else []
    if meta["w"] and meta["h"]:
        W0, H0 = int(meta["w"]), int(meta["h"])
    else:
        from PIL import Image
        with Image.open(paths[0]) as im:
            W0, H0 = im.size
    if strided:                 # rows / columns 0, f, 2f, ...: ceil(n / f) of them
        W, H = -(-W0 // downsample), -(-H0 // downsample)
        sx = sy = 1.0 / downsample
    else:
        W, H = W0 // downsample, H0 // downsample
        sx, sy = W / W0, H / H0
    if meta["fl_x"]:
        fx = float(meta["fl_x"]) * sx
        fy = float(meta["fl_y"] or meta["fl_x"]) * sy
    else:
        fx = fy = 0.5 * W / math.tan(0.5 * float(meta["camera_angle_x"]))
    cx = float(meta["cx"]) * sx if meta["cx"] is not None else W * 0.5
    cy = float(meta["cy"]) * sy if meta["cy"] is not None else H * 0.5
    images = _stack([_read_image(p, (W, H), downsample if strided else 0) for p in paths], device) if load_images else None
    poses = np.stack([np.array(f["transform_matrix"], dtype=np.float32) for f in frames])
    n = len(frames)
    if holdout > 0:
        i_test = np.arange(n)[::holdout]
        i_train = np.setdiff1d(np.arange(n), i_test)
        i_val = i_test
    else:
        i_train, i_val = np.arange(n_main), np.arange(n_main, n)
        i_test = i_val
    return Scene(images, poses, H, W, fx, fy, cx, cy, near, far, ndc, white_background, i_train, i_val, i_test, paths)


def load_scene(cfg: dict, device="cuda") -> Scene:
    """``dataset`` block of an experiment file -> Scene."""
    d = cfg["dataset"]
    white = bool(cfg["nerf"]["train"].get("white_background", False))
    kind = d["type"]
    if kind == "blender":
        return load_blender(d["basedir"], bool(d.get("half_res", False)), int(d.get("testskip", 1)), float(d["near"]),
                            float(d["far"]), white, device)
    if kind in ("robot", "transforms"):
        path = d["basedir"] if d["basedir"].endswith(".json") else os.path.join(d["basedir"], "transforms.json")
        hold = int(d.get("llffhold", 0)) if kind == "robot" else 0
        return load_transforms(path, int(d.get("downsample_factor", 1)), float(d["near"]), float(d["far"]),
                               not bool(d.get("no_ndc", True)), hold, white, device,
                               val_path=path.replace(".json", "_val.json"), strided=(kind == "robot"))
    if kind == "forward_grid":
        return forward_grid_scene(d.get("teacher_weights"), int(d.get("grid", 5)), int(d.get("width", 1280)),
                                  int(d.get("height", 720)), device, holdout=int(d.get("llffhold", 8)),
                                  spacing=float(d.get("spacing", 0.25)))
    if kind == "teacher":
        return teacher_scene(d.get("teacher_weights"), int(d.get("train_views", 100)), int(d.get("val_views", 25)),
                             int(d.get("resolution", 400)), device, near=float(d["near"]), far=float(d["far"]))
    raise ValueError(f"dataset.type: unknown dataset type {kind!r}")


# ---------------------------------------------------------------------------------------------
# synthetic teacher scene
# ---------------------------------------------------------------------------------------------
LEGO_CAMERA_ANGLE_X = 0.6911112070083618


def teacher_pose(k: int, n: int, phase: float = 0.0) -> np.ndarray:
    """View k of n on the ring used for the synthetic scene: azimuth 360 (k + phase) / n,
    elevation -30 + 15 sin(2 pi k / n) degrees, radius 4."""
    return spherical_pose(360.0 * (k + phase) / n, -30.0 + 15.0 * math.sin(2 * math.pi * k / max(n, 1)), 4.0)


def teacher_scene(weights_npz: Optional[str], n_train: int, n_val: int, res: int, device, near: float = 2.0,
                  far: float = 6.0, phase: float = 0.0, num_coarse: int = 64, num_fine: int = 128) -> Scene:
    """Render ``n_train`` + ``n_val`` views of a trained 4x128 checkpoint (flat ``coarse`` / ``fine``
    vectors in an .npz) with the HIP path and quantise them to 8 bits, like photographs."""
    from .model import ARCH_LEGO_CKPT, NeRFModel
    from .render import render_rays
    if weights_npz is None:
        weights_npz = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                   "lego_lowres3_weights.npz")
    w = np.load(weights_npz)
    dev = torch.device(device)
    c = NeRFModel(ARCH_LEGO_CKPT, torch.from_numpy(w["coarse"]).to(dev))
    f = NeRFModel(ARCH_LEGO_CKPT, torch.from_numpy(w["fine"]).to(dev))
    focal = 0.5 * res / math.tan(0.5 * LEGO_CAMERA_ANGLE_X)
    poses = [teacher_pose(k, n_train, phase) for k in range(n_train)]
    poses += [teacher_pose(k, n_val, phase + 0.5) for k in range(n_val)]     # held-out: between the training azimuths
    scene = Scene(None, np.stack(poses), res, res, focal, focal, res * 0.5, res * 0.5, near, far, False, False,
                  np.arange(n_train), np.arange(n_train, n_train + n_val), np.arange(n_train, n_train + n_val))
    imgs = torch.empty(len(poses), res, res, 3, dtype=torch.uint8, device=dev)
    with torch.no_grad():
        for k in range(len(poses)):
            rb = scene.rays(k, dev)
            for s in range(0, rb.shape[0], 65536):
                rgb = render_rays(rb[s:s + 65536], c, f, num_coarse=num_coarse, num_fine=num_fine)[0]
                imgs[k].view(-1, 3)[s:s + 65536] = (rgb.clamp(0, 1) * 255.0 + 0.5).to(torch.uint8)
    scene.images = imgs
    return scene


def forward_grid_scene(weights_npz: Optional[str], grid: int, width: int, height: int, device, holdout: int = 8,
                       spacing: float = 0.25, depth: float = 4.0, num_coarse: int = 64, num_fine: int = 128) -> Scene:
    """Forward-facing synthetic capture: ``grid`` x ``grid`` cameras in the z = 0 plane, ``spacing``
    apart, all looking down -z (identity rotation), horizontal field of view of the lego scenes.
    The photographs are the trained lego checkpoint rendered by the HIP path with the object
    ``depth`` in front of the camera plane (ordinary rays, near 2 / far 6 around the object); the
    scene itself is the NDC configuration: near 0, far 1, NDC near plane 1, every ``holdout``-th
    view held out."""
    from .model import ARCH_LEGO_CKPT, NeRFModel
    from .render import render_rays
    if weights_npz is None:
        weights_npz = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                   "lego_lowres3_weights.npz")
    w = np.load(weights_npz)
    dev = torch.device(device)
    c = NeRFModel(ARCH_LEGO_CKPT, torch.from_numpy(w["coarse"]).to(dev))
    f = NeRFModel(ARCH_LEGO_CKPT, torch.from_numpy(w["fine"]).to(dev))
    focal = 0.5 * width / math.tan(0.5 * LEGO_CAMERA_ANGLE_X)
    poses = []
    for gy in range(grid):
        for gx in range(grid):
            m = np.eye(4, dtype=np.float32)
            m[0, 3] = (gx - 0.5 * (grid - 1)) * spacing
            m[1, 3] = (gy - 0.5 * (grid - 1)) * spacing
            poses.append(m)
    n = len(poses)
    i_test = np.arange(n)[::holdout] if holdout > 0 else np.zeros(0, np.int64)
    i_train = np.setdiff1d(np.arange(n), i_test)
    scene = Scene(None, np.stack(poses), height, width, focal, focal, width * 0.5, height * 0.5, 0.0, 1.0, True, False,
                  i_train, i_test, i_test)
    # the teacher sees the same cameras from its own frame: object at the origin, cameras at z = +depth,
    # lying on its side (the lego scenes are z-up; the cameras here look along -z)
    to_teacher = np.array([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float32)
    lift = np.eye(4, dtype=np.float32); lift[2, 3] = depth
    photo = Scene(None, np.stack([to_teacher @ lift @ p for p in poses]), height, width, focal, focal, width * 0.5,
                  height * 0.5, depth - 2.0, depth + 2.0, False, False)
    imgs = torch.empty(n, height, width, 3, dtype=torch.uint8, device=dev)
    with torch.no_grad():
        for k in range(n):
            rb = photo.rays(k, dev)
            for s in range(0, rb.shape[0], 131072):
                rgb = render_rays(rb[s:s + 131072], c, f, num_coarse=num_coarse, num_fine=num_fine)[0]
                imgs[k].view(-1, 3)[s:s + 131072] = (rgb.clamp(0, 1) * 255.0 + 0.5).to(torch.uint8)
    scene.images = imgs
    return scene


# ---------------------------------------------------------------------------------------------
# per-iteration ray sampler
# ---------------------------------------------------------------------------------------------
class RaySampler:
    """One training view per iteration, ``rays`` pixels of it drawn on the GPU.

    The view index and the pixels are functions of (seed, iteration, rank) only, so a run is
    reproducible and every rank of a data-parallel job draws different rays."""

    def __init__(self, scene: Scene, rays: int, seed: int = 42, rank: int = 0, world: int = 1):
        if scene.images is None or len(scene.i_train) == 0:
            raise ValueError("RaySampler needs a scene with images and a non-empty training split")
        if scene.images.dtype != torch.uint8 or not scene.images.is_cuda or not scene.images.is_contiguous():
            raise ValueError("scene.images must be a contiguous uint8 GPU tensor [N, H, W, C]")
        self.scene, self.R, self.seed, self.rank, self.world = scene, int(rays), int(seed) & _M32, int(rank), int(world)
        dev = scene.images.device
        self.ray_batch = torch.empty(self.R, 11, dtype=torch.float32, device=dev)
        self.target = torch.empty(self.R, 3, dtype=torch.float32, device=dev)
        self.pixels = torch.empty(self.R, dtype=torch.int32, device=dev)
        self._cams = {int(k): scene.camera(int(k)) for k in scene.i_train}
        self._L = _lib.load()

    def view_of(self, iteration: int) -> int:
        draw = rand_u32(self.seed ^ 0x5EED5EED, _lib.STREAM_PIXELS, iteration * self.world + self.rank)
        return int(self.scene.i_train[(draw * len(self.scene.i_train)) >> 32])

    def counter_of(self, iteration: int) -> int:
        return (iteration * self.world + self.rank) * self.R

    def next(self, iteration: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(ray_batch [R, 11], target [R, 3]) of this iteration; the buffers are reused by the next call."""
        from .render import _stream
        k = self.view_of(iteration)
        img = self.scene.images[k]
        dev = img.device
        with torch.cuda.device(dev):
            _lib.check(self._L.nk_sample_training_rays(
                self._cams[k], img.data_ptr(), int(img.shape[-1]), int(self.scene.white_background), self.seed,
                self.counter_of(iteration), self.R, self.ray_batch.data_ptr(), self.target.data_ptr(),
                self.pixels.data_ptr(), _stream(dev)))
        return self.ray_batch, self.target
