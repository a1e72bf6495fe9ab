"""Image and capture-quality metrics the reference reports.

``psnr_8bit``      PSNR between two 8-bit images exactly as the reference's evaluation cell computes
                   it (/root/reference/instant_ngp_collab_cloud.ipynb:1550-1568): both images to
                   float32, ``mse = mean((a - b)^2)`` over all pixels and channels,
                   ``20 log10(255 / sqrt(mse))``, ``inf`` for identical images.
``mse2psnr``       ``-10 log10(mse)`` on [0, 1] colours: the ``psnr`` scalar stored in the
                   reference's checkpoints next to ``loss``.
``analyze_poses``  camera-pose statistics of a transforms.json
                   (/root/reference/paralaje.ipynb:41-74): positions centred and scaled into
                   [-1, 1], their distance statistics, and the spread of viewing directions --
                   the low-parallax indicator of the robot captures.
"""
from __future__ import annotations

import json
import math
from typing import Dict, Sequence

import numpy as np


def psnr_8bit(img_a, img_b) -> float:
    a = np.asarray(img_a)
    b = np.asarray(img_b)
    if a.shape != b.shape:
        raise ValueError(f"images must have the same size, got {a.shape} and {b.shape}")
    a = a.astype(np.float32)
    b = b.astype(np.float32)
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float("inf")
    return float(20 * np.log10(255.0 / np.sqrt(mse)))


def mse2psnr(mse: float) -> float:
    return -10.0 * math.log10(max(float(mse), 1e-12))


def to_8bit(rgb) -> np.ndarray:
    """[..., 3] float colours in [0, 1] -> uint8 (round to nearest)."""
    return (np.clip(np.asarray(rgb, dtype=np.float32), 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)


def analyze_poses(frames: Sequence[dict], label: str = "") -> Dict:
    """Statistics of the camera poses of ``frames`` (each with a 4x4 ``transform_matrix``)."""
    mats = [np.array(f["transform_matrix"], dtype=np.float64) for f in frames]
    positions = np.array([m[:3, 3] for m in mats])
    forward = np.array([-m[:3, 2] for m in mats])
    positions = positions - positions.mean(axis=0)
    span = np.abs(positions).max()
    if span > 0:
        positions = positions / span
    dist = np.linalg.norm(positions, axis=1)
    return {"label": label, "positions": positions, "forward_vectors": forward,
            "metrics": {"dataset": label, "num_frames": len(mats), "mean_distance": float(np.mean(dist)),
                        "spatial_std": float(np.std(dist)), "spatial_range": float(np.max(dist) - np.min(dist)),
                        "directional_std": float(np.std(forward, axis=0).mean())}}


def main(argv=None) -> None:
    import argparse
    ap = argparse.ArgumentParser(description="Camera-pose statistics of transforms.json files")
    ap.add_argument("transforms", nargs="+", help="transforms.json files")
    a = ap.parse_args(argv)
    rows = []
    for path in a.transforms:
        with open(path) as fh:
            frames = [f for f in json.load(fh)["frames"] if "transform_matrix" in f]
        rows.append(analyze_poses(frames, path)["metrics"])
    keys = ["dataset", "num_frames", "mean_distance", "spatial_std", "spatial_range", "directional_std"]
    print("  ".join(f"{k:>16s}" for k in keys))
    for r in rows:
        print("  ".join(f"{r[k]:>16.6f}" if isinstance(r[k], float) else f"{str(r[k])[-16:]:>16s}" for k in keys))


if __name__ == "__main__":
    main()
