#!/usr/bin/env python3
"""poses.txt (robot forward-kinematics camera poses) -> transforms.json family.

Drop-in for the reference's converter (/root/reference/parser_instant_ngp.py): same command
line (:174-190), same four output files (:297-355) and the same numbers, so robot-FK captures
feed the trainer unchanged.  Behaviour kept on purpose, including its quirks:

* matrices are ``[a, b, c, d ; ... ]`` blocks, rows split on ';', four numbers per row, only
  4-row blocks are kept (:20-46);
* the FIRST matrix / first existing image is the validation frame, the rest train (:196-202);
* ``--recenter`` shifts by the centroid of the TRAINING cameras, scales translations by
  ``scale_trans / max distance`` and then negates the third rotation column and t_z, which makes
  det(R) = -1 (:209-239); without ``--recenter`` nothing is changed although a Z inversion is
  announced (:242-244);
* intrinsics come from the field of view (default 87 x 58 degrees, the RealSense D405) (:266-272);
* training frames take the next existing ``"{prefix}{idx}.{ext}"`` starting at idx 1, skipping
  gaps (:278-295); per-frame ``sharpness`` is the variance of a 3x3 Laplacian of the grey image
  (:49-60);
* ``aabb_scale`` is the smallest power of two >= the camera-cloud diameter, capped at 128
  (:148-171);
* ``_test.json`` holds 8 poses on a circle of radius 50 around the scene centre (key
  ``transform_matrix_start``), ``_test_video.json`` 60 poses on radius 40 with a 5 sin(4 theta)
  height wobble (:63-125, :328-355).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import re
from typing import Dict, List, Optional, Sequence

import numpy as np
from PIL import Image, ImageFilter

_BLOCK = re.compile(r"\[\s*(.*?)\s*\]", re.DOTALL)
_NUMBER = re.compile(r"-?\d+\.?\d*(?:[eE][-+]?\d+)?")


def read_pose_matrices(path: str) -> List[List[List[float]]]:
    """All 4x4 matrices of a poses.txt, as nested Python lists."""
    with open(path, "r") as fh:
        text = fh.read()
    out = []
    for block in _BLOCK.findall(text):
        rows = []
        for line in block.split(";"):
            line = line.strip().replace(",", " ")
            if not line:
                continue
            vals = _NUMBER.findall(line)
            if len(vals) == 4:
                rows.append([float(v) for v in vals])
        if len(rows) == 4:
            out.append(np.array(rows).tolist())
    if not out:
        raise RuntimeError(f"No matrices parsed from {path}")
    return out


def laplacian_sharpness(image_path: str) -> float:
    """Variance of the 3x3 Laplacian response of the 8-bit grey image (PIL only)."""
    grey = Image.open(image_path).convert("L")
    lap = grey.filter(ImageFilter.Kernel(size=(3, 3), kernel=[0, 1, 0, 1, -4, 1, 0, 1, 0], scale=1, offset=0))
    return float(np.array(lap, dtype=np.float64).var())


def camera_positions(mats: Sequence) -> np.ndarray:
    return np.array([[m[0][3], m[1][3], m[2][3]] for m in mats])


def aabb_scale_for(mats: Sequence, margin: float = 1.0, verbose: bool = True) -> float:
    pos = camera_positions(mats)
    centre = pos.mean(axis=0)
    reach = np.linalg.norm(pos - centre, axis=1).max()
    if verbose:
        print(f"Scene center: {centre}")
        print(f"Max distance from center: {reach}")
    want = reach * 2 * margin
    scale = 1.0
    while scale < want and scale < 128.0:
        scale *= 2.0
    return scale


def ring_poses(centre, radius: float, count: int, wobble: float = 0.0) -> List[list]:
    """Cameras on a horizontal circle around ``centre`` looking at it (z up); ``wobble`` adds
    ``wobble * sin(4 theta)`` to the height."""
    poses = []
    for k in range(count):
        theta = 2 * math.pi * k / count
        x = centre[0] + radius * math.cos(theta)
        y = centre[1] + radius * math.sin(theta)
        z = centre[2] + (wobble * math.sin(4 * theta) if wobble else 0)
        fwd = np.array([centre[0] - x, centre[1] - y, centre[2] - z])
        fwd = fwd / np.linalg.norm(fwd)
        right = np.cross(fwd, np.array([0, 0, 1]))
        right = right / np.linalg.norm(right)
        up = np.cross(right, fwd)
        poses.append([[right[0], up[0], -fwd[0], x],
                      [right[1], up[1], -fwd[1], y],
                      [right[2], up[2], -fwd[2], z],
                      [0.0, 0.0, 0.0, 1.0]])
    return poses


def _header(w, h, cax, cay, fl_x, fl_y, cx, cy, k1, k2, p1, p2, aabb) -> Dict:
    return {"camera_angle_x": cax, "camera_angle_y": cay, "fl_x": fl_x, "fl_y": fl_y, "k1": k1, "k2": k2,
            "p1": p1, "p2": p2, "cx": cx, "cy": cy, "w": w, "h": h, "aabb_scale": aabb, "frames": []}


def _recenter(mat, centre, factor) -> None:
    for r in range(3):
        mat[r][3] = (mat[r][3] - centre[r]) * factor
    for r in range(3):
        mat[r][2] *= -1
    mat[2][3] *= -1


def convert(poses: str, image_folder: str, image_prefix: str = "TestNERF ", image_ext: str = "jpg",
            camera_angle_x: float = 87.0, camera_angle_y: float = 58.0, k1: float = 0.0, k2: float = 0.0,
            p1: float = 0.0, p2: float = 0.0, recenter: bool = False, scale_trans: float = 1.0,
            verbose: bool = True) -> Dict[str, Dict]:
    """The four JSON documents, keyed ``train / val / test / test_video``."""
    say = print if verbose else (lambda *a, **k: None)
    mats = read_pose_matrices(poses)
    if len(mats) <= 1:
        raise RuntimeError("Need at least 2 matrices to separate training and validation")
    val, train = mats[0], mats[1:]
    say(f"Initial aabb_scale (before transformations): {aabb_scale_for(train, verbose=verbose)}")
    if recenter:
        pos = camera_positions(train)
        centre = pos.mean(axis=0)
        factor = scale_trans / np.linalg.norm(pos - centre, axis=1).max()
        for m in train:
            _recenter(m, centre, factor)
        _recenter(val, centre, factor)
        say(f"Applied recentering: {recenter}, scale factor: {factor}")
    else:
        say("Applied Z-axis inversion for Instant-NGP compatibility")   # (announced only, as in the reference)
    aabb = aabb_scale_for(train, verbose=verbose)
    say(f"Final aabb_scale (after transformations): {aabb}")
    scene_centre = camera_positions(train).mean(axis=0)

    def image(idx):
        return os.path.join(image_folder, f"{image_prefix}{idx}.{image_ext}")

    size = None
    for idx in range(len(train) * 2):
        if os.path.isfile(image(idx)):
            with Image.open(image(idx)) as im:
                size = im.size
            break
    if size is None:
        raise FileNotFoundError("No images found matching prefix and extension in image_folder")
    w, h = size
    cax, cay = math.radians(camera_angle_x), math.radians(camera_angle_y)
    head = (w, h, cax, cay, 0.5 * w / math.tan(cax / 2), 0.5 * h / math.tan(cay / 2), w / 2.0, h / 2.0,
            k1, k2, p1, p2, aabb)

    docs = {name: _header(*head) for name in ("train", "val", "test", "test_video")}
    nxt = 1
    for m in train:
        placed = False
        while nxt < len(mats) * 2:
            path = image(nxt)
            nxt += 1
            if os.path.isfile(path):
                docs["train"]["frames"].append({"file_path": path, "sharpness": laplacian_sharpness(path),
                                                "transform_matrix": m})
                placed = True
                break
        if not placed:
            raise FileNotFoundError(f"No image found for pose, checked up to index {nxt - 1}.")
    for idx in range(len(mats) * 2):
        if os.path.isfile(image(idx)):
            docs["val"]["frames"].append({"file_path": image(idx), "sharpness": laplacian_sharpness(image(idx)),
                                          "transform_matrix": val})
            break
    else:
        raise FileNotFoundError("No image found for validation.")
    for k, pose in enumerate(ring_poses(scene_centre, 50, 8)):
        docs["test"]["frames"].append({"file_path": f"./Test{k}.jpg", "transform_matrix_start": pose})
    for pose in ring_poses(scene_centre, 40, 60, wobble=5):
        docs["test_video"]["frames"].append({"transform_matrix": pose})
    return docs


def write(docs: Dict[str, Dict], output: str, verbose: bool = True) -> List[str]:
    names = {"train": output, "val": output.replace(".json", "_val.json"),
             "test": output.replace(".json", "_test.json"), "test_video": output.replace(".json", "_test_video.json")}
    for key, path in names.items():
        with open(path, "w") as fh:
            json.dump(docs[key], fh, indent=2)
        if verbose:
            print(f"Wrote {path} with {len(docs[key]['frames'])} frames.")
    return list(names.values())


def main(argv: Optional[Sequence[str]] = None) -> None:
    ap = argparse.ArgumentParser(description="Convert poses.txt to Instant-NGP transforms.json with sharpness")
    ap.add_argument("--poses", default="poses.txt", help="Path to poses.txt")
    ap.add_argument("--image_folder", default="images_robot", help="Folder with images")
    ap.add_argument("--image_prefix", default="TestNERF ", help="Image name prefix")
    ap.add_argument("--image_ext", default="jpg", help="Image file extension")
    ap.add_argument("--camera_angle_x", type=float, default=87.0, help="Horizontal FOV (deg)")
    ap.add_argument("--camera_angle_y", type=float, default=58.0, help="Vertical FOV (deg)")
    for name in ("k1", "k2", "p1", "p2"):
        ap.add_argument(f"--{name}", type=float, default=0.0, help=f"Distortion {name}")
    ap.add_argument("--recenter", action="store_true", help="Recenter cameras to scene origin")
    ap.add_argument("--scale_trans", type=float, default=1.0, help="Uniform scale for camera translations")
    ap.add_argument("--output", default="transforms.json", help="Output JSON name")
    a = ap.parse_args(argv)
    docs = convert(a.poses, a.image_folder, a.image_prefix, a.image_ext, a.camera_angle_x, a.camera_angle_y,
                   a.k1, a.k2, a.p1, a.p2, a.recenter, a.scale_trans)
    write(docs, a.output)


if __name__ == "__main__":
    main()
