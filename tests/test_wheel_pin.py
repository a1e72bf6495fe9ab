"""What the reference's own nerf-pytorch result for the wheel capture pins (CPU).

tests/golden/wheel_ref_target_160x90.png and wheel_ref_render_160x90.png are the exact 160 x 90
validation target and render behind the reference's screenshots
(/root/reference/results/wheel/nerf-pytorch/imgs/{reals,rendered}/TestNERF_0_original*.png; made by
tools/make_wheel_pin.py).  They pin three things about the un-vendored ``robot`` loader:
  * the validation view is the FIRST pose / ``TestNERF 0.jpg`` (what ``llffhold: 8`` holds out first);
  * images are reduced by striding, ``img[::8, ::8]``, not by a filtered resize;
  * the logged PSNR is the float one (-10 log10 mse), and this pair sits at 9.71 dB.
The camera convention that produced the render is NOT pinned: tools/pin_wheel_search.py lists the
bounded grid that was searched without reproducing it (profiles/r02_wheel_pin_search.txt).
"""
import glob
import os

import numpy as np
from PIL import Image


def _img(path):
    return np.asarray(Image.open(path).convert("RGB")).astype(np.float64) / 255.0


def _psnr(a, b):
    return -10.0 * np.log10(((a - b) ** 2).mean())


def test_validation_view_and_downsampling_rule(golden_dir):
    target = _img(os.path.join(golden_dir, "wheel_ref_target_160x90.png"))
    assert target.shape == (90, 160, 3)
    scores = {}
    for p in glob.glob(os.path.join(golden_dir, "wheel_real_robot", "images_robot", "*.jpg")):
        k = int(os.path.basename(p).split()[1].split(".")[0])
        scores[k] = _psnr(_img(p)[::8, ::8], target)
    assert max(scores, key=scores.get) == 0                 # the first frame is the held-out view
    assert scores[0] > 40.0                                 # equal up to the JPEG decoder
    assert sorted(scores.values())[-2] < 15.0               # no other photograph comes close
    photo = Image.open(os.path.join(golden_dir, "wheel_real_robot", "images_robot", "TestNERF 0.jpg")).convert("RGB")
    for filt in (Image.BOX, Image.BILINEAR, Image.BICUBIC, Image.LANCZOS):
        assert _psnr(np.asarray(photo.resize((160, 90), filt)).astype(np.float64) / 255.0, target) < 25.0
    for oy, ox in ((1, 0), (0, 1), (4, 4)):                 # and the stride starts at pixel 0
        shifted = _img(os.path.join(golden_dir, "wheel_real_robot", "images_robot", "TestNERF 0.jpg"))[oy::8, ox::8]
        assert shifted.shape == target.shape and _psnr(shifted, target) < 30.0


def test_robot_loader_reproduces_the_reference_target(golden_dir):
    """The product loader's image reduction for ``dataset.type: robot`` gives the reference's target."""
    from nerf_kinematics_amd import datasets
    target = _img(os.path.join(golden_dir, "wheel_ref_target_160x90.png"))
    small = datasets._read_image(os.path.join(golden_dir, "wheel_real_robot", "images_robot", "TestNERF 0.jpg"), stride=8)
    assert small.shape == (90, 160, 3) and small.dtype == np.uint8
    assert _psnr(small.astype(np.float64) / 255.0, target) > 40.0


def test_reference_render_psnr_recorded(golden_dir):
    """The reference's render of that view against its target: 9.71 dB (the checkpoint's stored
    ``psnr`` 8.82 = -10 log10(0.1312) belongs to another validation draw: i_val = [0, 8] and the
    validation view is drawn at random)."""
    from nerf_kinematics_amd.metrics import mse2psnr, psnr_8bit
    target = _img(os.path.join(golden_dir, "wheel_ref_target_160x90.png"))
    render = _img(os.path.join(golden_dir, "wheel_ref_render_160x90.png"))
    mse = float(((render - target) ** 2).mean())
    assert abs(mse - 0.10692) < 1e-4 and abs(mse2psnr(mse) - 9.7096) < 1e-3
    w = np.load(os.path.join(golden_dir, "wheel_robot6_weights.npz"))
    assert abs(float(w["psnr"]) - 8.8197) < 1e-3 and abs(mse2psnr(float(w["loss"])) - float(w["psnr"])) < 1e-3
    assert abs(float(w["psnr"]) - 9.7096) > 0.5
    assert abs(psnr_8bit((render * 255).astype(np.uint8), (target * 255).astype(np.uint8)) - 9.7096) < 0.05
