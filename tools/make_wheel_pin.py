"""Fixtures from the reference's own nerf-pytorch result for the wheel capture (build container only:
reads /root/reference, writes tests/golden/).

/root/reference/results/wheel/nerf-pytorch/imgs/rendered/TestNERF_0_original_gen.png (636 x 356) and
.../imgs/reals/TestNERF_0_original.png (640 x 358) are screenshots of a 160 x 90 render and of its
160 x 90 target, enlarged about four times with nearest-neighbour sampling: every source pixel is a
constant block.  The block edges are found from the data (columns / rows where anything changes)
and one sample per block gives back the exact 160 x 90 images (in-block deviation 0).
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/results/wheel/nerf-pytorch/imgs"


def unscale(path):
    g = np.asarray(Image.open(path).convert("RGB"))
    f = g.astype(np.int64)
    bx = np.nonzero(np.abs(np.diff(f, axis=1)).sum(axis=(0, 2)) > 0)[0]
    by = np.nonzero(np.abs(np.diff(f, axis=0)).sum(axis=(1, 2)) > 0)[0]
    xs, ys = np.concatenate([[0], bx + 1]), np.concatenate([[0], by + 1])
    xe, ye = np.concatenate([bx + 1, [g.shape[1]]]), np.concatenate([by + 1, [g.shape[0]]])
    out = g[ys][:, xs]
    dev = max(int(np.ptp(g[ys[j]:ye[j], xs[k]:xe[k]].reshape(-1, 3), axis=0).max())
              for j in range(len(ys)) for k in range(len(xs)))
    assert out.shape == (90, 160, 3) and dev == 0, (out.shape, dev)
    return out


if __name__ == "__main__":
    gen = unscale(os.path.join(REF, "rendered", "TestNERF_0_original_gen.png"))
    real = unscale(os.path.join(REF, "reals", "TestNERF_0_original.png"))
    Image.fromarray(gen).save(os.path.join(ROOT, "tests", "golden", "wheel_ref_render_160x90.png"))
    Image.fromarray(real).save(os.path.join(ROOT, "tests", "golden", "wheel_ref_target_160x90.png"))
    mse = float(((gen / 255.0 - real / 255.0) ** 2).mean())
    print(f"render vs target: mse {mse:.5f}, psnr {-10 * np.log10(mse):.4f} dB")
