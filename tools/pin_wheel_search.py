"""Bounded search (CPU, oracle only) for the camera convention under which the reference's trained
wheel_robot6 networks reproduce the reference's own 160 x 90 validation render
(tests/golden/wheel_ref_render_160x90.png, from tools/make_wheel_pin.py).

The reference's ``robot`` loader is not vendored, so what it did to the robot's poses is open.
Candidates = (axis convention of the camera matrix) x (pose normalisation) x (order of the two) x
(translation scale) x (focal length), all rendered through the oracle's get_rays -> ndc_rays ->
render_rays with the checkpoint's weights at the validation pose (first matrix of poses.txt =
``TestNERF 0.jpg``), 64 + 128 samples, every 4th pixel.  Score: correlation and PSNR against the
reference render; and, for pose 1, the best PSNR against any of the twelve photographs (a
convention that matches training would reproduce a training photograph at ~19 dB, the run's
training PSNR).

    python tools/pin_wheel_search.py [--quick] > profiles/r02_wheel_pin_search.txt
"""
import argparse
import glob
import itertools
import os
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nerf_oracle as O                      # noqa: E402
from nerf_kinematics_amd import poses as P               # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
H, W, S = 90, 160, 4
FX, FY = 674.4192801798159 / 8, 649.4571918977125 / 8      # converter intrinsics / downsample_factor

AXES = ["none", "cv2gl", "parser", "llff", "negz", "negyz_t"]
NORMS = ["raw", "m", "parser_rc", "rc", "rc_m", "rc_unit"]


def axes(ps, kind):
    ps = ps.copy()
    if kind == "cv2gl":        # OpenCV -> OpenGL camera axes: y and z columns negated
        ps[:, :3, 1] *= -1; ps[:, :3, 2] *= -1
    elif kind == "parser":     # what the pose converter does with --recenter: z column and t_z negated
        ps[:, :3, 2] *= -1; ps[:, 2, 3] *= -1
    elif kind == "llff":       # LLFF loader's column swap [y, -x, z]
        q = ps.copy(); q[:, :3, 0] = ps[:, :3, 1]; q[:, :3, 1] = -ps[:, :3, 0]; ps = q
    elif kind == "negz":
        ps[:, :3, 2] *= -1
    elif kind == "negyz_t":
        ps[:, :3, 1] *= -1; ps[:, :3, 2] *= -1; ps[:, 1, 3] *= -1; ps[:, 2, 3] *= -1
    return ps


def recenter(ps):              # LLFF recenter_poses: inverse of the average pose applied to all
    n = lambda v: v / np.linalg.norm(v)
    c = ps[:, :3, 3].mean(0); z = n(ps[:, :3, 2].sum(0)); up = ps[:, :3, 1].sum(0)
    x = n(np.cross(up, z)); y = n(np.cross(z, x))
    avg = np.eye(4); avg[:3, 0], avg[:3, 1], avg[:3, 2], avg[:3, 3] = x, y, z, c
    return np.linalg.inv(avg) @ ps


def norm(ps, kind):
    ps = ps.copy()
    if kind == "m":            # millimetres -> metres
        ps[:, :3, 3] /= 1000
    elif kind == "parser_rc":  # converter's --recenter: centroid / max distance of the 11 training poses
        t = ps[1:, :3, 3]; c = t.mean(0); ps[:, :3, 3] = (ps[:, :3, 3] - c) / np.linalg.norm(t - c, axis=1).max()
    elif kind == "rc":
        ps = recenter(ps)
    elif kind == "rc_m":
        ps[:, :3, 3] /= 1000; ps = recenter(ps)
    elif kind == "rc_unit":
        ps = recenter(ps); ps[:, :3, 3] /= np.linalg.norm(ps[:, :3, 3], axis=1).max()
    return ps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="first grid only")
    a = ap.parse_args()
    torch.set_num_threads(max(1, (os.cpu_count() or 2) - 2))
    w = np.load(os.path.join(G, "wheel_robot6_weights.npz"))
    co = O.NeRFNet.from_flat(O.ARCH_WHEEL_CKPT, torch.from_numpy(w["coarse"]))
    fo = O.NeRFNet.from_flat(O.ARCH_WHEEL_CKPT, torch.from_numpy(w["fine"]))
    gen = np.asarray(Image.open(os.path.join(G, "wheel_ref_render_160x90.png"))).astype(np.float32) / 255
    photos = {}
    for p in glob.glob(os.path.join(G, "wheel_real_robot", "images_robot", "*.jpg")):
        photos[int(os.path.basename(p).split()[1].split(".")[0])] = \
            np.asarray(Image.open(p).convert("RGB"))[::8, ::8].astype(np.float32) / 255
    raw = np.array(P.read_pose_matrices(os.path.join(G, "wheel_real_robot", "poses.txt")))
    sl = (slice(None, None, S), slice(None, None, S))

    def render(c2w, focal):
        c2w = torch.as_tensor(np.asarray(c2w)[:3, :4], dtype=torch.float32)
        ro, rd = O.get_rays(H, W, focal, c2w)
        vd = rd / rd.norm(dim=-1, keepdim=True)
        ro, rd = O.ndc_rays(H, W, focal, 1.0, ro, rd)
        ro, rd, vd = ro[sl].reshape(-1, 3), rd[sl].reshape(-1, 3), vd[sl].reshape(-1, 3)
        n = ro.shape[0]
        rb = torch.cat([ro, rd, torch.zeros(n, 1), torch.ones(n, 1), vd], -1)
        with torch.no_grad():
            return np.nan_to_num(O.render_rays(rb, co, fo, O.RenderOpts(64, 128))["rgb_fine"].numpy())

    def score(ps, focal):
        a0 = render(ps[0], focal); b = gen[sl].reshape(-1, 3)
        c = float(np.corrcoef(a0.ravel(), b.ravel())[0, 1]) if a0.std() > 0 else float("nan")
        p = float(-10 * np.log10(((a0 - b) ** 2).mean()))
        a1 = render(ps[1], focal)
        tr = max((float(-10 * np.log10(((a1 - photos[k][sl].reshape(-1, 3)) ** 2).mean())), k) for k in photos)
        return c, p, tr

    rows = []
    print("# grid 1: axes x normalisation x order x focal (fl_x/8, fl_y/8)")
    for ax, nm, order in itertools.product(AXES, NORMS, ("axes-first", "norm-first")):
        ps = norm(axes(raw, ax), nm) if order == "axes-first" else axes(norm(raw, nm), ax)
        for focal in (FX, FY):
            c, p, tr = score(ps, focal)
            rows.append((c, p, tr[0], f"{ax} {nm} {order} f={focal:.2f}"))
            print(f"corr {c:+.3f} psnr {p:5.2f} | pose 1 vs photo {tr[1]:2d}: {tr[0]:5.2f} dB | {rows[-1][3]}", flush=True)
    if not a.quick:
        print("# grid 2: axes x (LLFF recentring on / off) x translation scale 1e-4 .. 316 x focal")
        for ax, rc in itertools.product(("none", "cv2gl", "parser", "llff"), (True, False)):
            for focal in (20, 40, 60, FX, 120, 170, 340, 674.4):
                for s in np.logspace(-4, 2.5, 14):
                    ps = axes(raw, ax).copy(); ps[:, :3, 3] *= s
                    if rc:
                        ps = recenter(ps)
                    c, p, tr = score(ps, focal)
                    rows.append((c, p, tr[0], f"{ax} recenter={rc} scale={s:.5f} f={focal:.1f}"))
                    print(f"corr {c:+.3f} psnr {p:5.2f} | pose 1 vs photo {tr[1]:2d}: {tr[0]:5.2f} dB | {rows[-1][3]}", flush=True)
        print("# grid 3: every signed permutation of the camera axes (48) x pose as given / inverted x LLFF recentring x 5 scales, f = fl_x/8")
        for perm in itertools.permutations(range(3)):
            for signs in itertools.product((1, -1), repeat=3):
                b = np.zeros((3, 3))
                for j, (pj, sg) in enumerate(zip(perm, signs)):
                    b[pj, j] = sg                      # new camera axis j = sg * old axis pj
                for inv in (False, True):
                    base = np.linalg.inv(raw) if inv else raw.copy()
                    ps0 = base.copy(); ps0[:, :3, :3] = base[:, :3, :3] @ b
                    for s in (0.001, 0.03, 0.3, 1.0, 3.0):
                        ps = ps0.copy(); ps[:, :3, 3] *= s; ps = recenter(ps)
                        c, p, tr = score(ps, FX)
                        rows.append((c, p, tr[0], f"axes {perm} signs {signs} inverted={inv} scale={s}"))
                        print(f"corr {c:+.3f} psnr {p:5.2f} | pose 1 vs photo {tr[1]:2d}: {tr[0]:5.2f} dB | {rows[-1][3]}", flush=True)
    rows = [r for r in rows if r[0] == r[0]]
    print("# best by correlation with the reference render")
    for r in sorted(rows)[-8:]:
        print(f"corr {r[0]:+.3f} psnr {r[1]:5.2f} train {r[2]:5.2f} | {r[3]}")
    print("# best by PSNR of pose 1 against a photograph")
    for r in sorted(rows, key=lambda r: r[2])[-8:]:
        print(f"corr {r[0]:+.3f} psnr {r[1]:5.2f} train {r[2]:5.2f} | {r[3]}")


if __name__ == "__main__":
    main()
