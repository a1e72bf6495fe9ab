#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ (run in the build container, where
/root/reference is mounted; the GPU box only ever sees the fixtures).

1. Weights of the two vendored nerf-pytorch checkpoints, flattened in state-dict order, plus the
   scalars stored beside them (iter, loss, psnr, lr).  Loaded with ``weights_only=True``.
2. Golden vectors of the oracle on the lego checkpoint: per-stage intermediates for 64 probe rays
   (deterministic and perturbed), and the 100x100 probe render statistics.
3. The four JSON files the reference's poses.txt converter writes for the wheel dataset
   (produced by importing and running the reference's own parser), as loader fixtures.
"""
import json
import os
import runpy
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nerf_oracle as O  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


def dump_ckpt(name, arch, out_name):
    ck = torch.load(f"{REF}/trained_models/nerf-pytorch/{name}/checkpoint110000.ckpt",
                    weights_only=True, map_location="cpu")
    coarse = O.NeRFNet.flatten_state_dict(arch, ck["model_coarse_state_dict"]).numpy()
    fine = O.NeRFNet.flatten_state_dict(arch, ck["model_fine_state_dict"]).numpy()
    og = ck["optimizer_state_dict"]
    np.savez(os.path.join(GOLD, out_name), coarse=coarse, fine=fine,
             iter=np.int64(ck["iter"]), loss=np.float32(float(ck["loss"])),
             psnr=np.float64(ck["psnr"]), lr=np.float64(og["param_groups"][0]["lr"]),
             adam_step=np.float32(float(og["state"][0]["step"])))
    print(out_name, coarse.shape, fine.shape)
    return coarse, fine


def dump_layout(name, out_name):
    """Structure of a reference checkpoint (keys, shapes, dtypes, optimiser bookkeeping): what a
    writer has to reproduce for the file to load back into upstream code."""
    ck = torch.load(f"{REF}/trained_models/nerf-pytorch/{name}/checkpoint110000.ckpt",
                    weights_only=True, map_location="cpu")
    desc = lambda t: [list(t.shape), str(t.dtype)]
    og = ck["optimizer_state_dict"]
    layout = {
        "top_level": {k: type(v).__name__ for k, v in ck.items()},
        "iter": int(ck["iter"]), "psnr": float(ck["psnr"]), "loss": desc(ck["loss"]),
        "model_coarse_state_dict": {k: desc(v) for k, v in ck["model_coarse_state_dict"].items()},
        "model_fine_state_dict": {k: desc(v) for k, v in ck["model_fine_state_dict"].items()},
        "optimizer_state": {str(i): {k: desc(v) for k, v in e.items()} for i, e in og["state"].items()},
        "optimizer_step": float(og["state"][0]["step"]),
        "param_groups": [{k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in g.items()}
                         for g in og["param_groups"]],
    }
    with open(os.path.join(GOLD, out_name), "w") as fh:
        json.dump(layout, fh, indent=1)
    print(out_name)


def probe_rays(H=100, W=100):
    focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
    c2w = O.spherical_pose(45.0, -30.0, 4.0)
    ro, rd = O.get_rays(H, W, float(focal), c2w)
    return O.make_ray_batch(ro, rd, 2.0, 6.0), c2w, focal


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    coarse, fine = dump_ckpt("lego-lowres3", O.ARCH_LEGO_CKPT, "lego_lowres3_weights.npz")
    dump_ckpt("wheel_robot6", O.ARCH_WHEEL_CKPT, "wheel_robot6_weights.npz")
    dump_layout("lego-lowres3", "lego_ckpt_layout.json")
    dump_layout("wheel_robot6", "wheel_ckpt_layout.json")

    # ---- probe render (oracle pin) ----------------------------------------------------------
    net_c = O.NeRFNet.from_flat(O.ARCH_LEGO_CKPT, torch.from_numpy(coarse))
    net_f = O.NeRFNet.from_flat(O.ARCH_LEGO_CKPT, torch.from_numpy(fine))
    rb, c2w, focal = probe_rays()
    opts = O.RenderOpts(num_coarse=64, num_fine=128)
    with torch.no_grad():
        out = O.render_rays(rb, net_c, net_f, opts)
    rgb = out["rgb_fine"].view(100, 100, 3).numpy()
    acc = out["acc_fine"].view(100, 100).numpy()
    stats = {"mean_rgb": rgb.reshape(-1, 3).mean(0).tolist(),
             "centre_acc": float(acc[40:60, 40:60].mean()),
             "corner_acc": float(acc[:10, :10].mean())}
    print("probe:", stats)
    from PIL import Image
    Image.fromarray((np.clip(rgb, 0, 1) * 255).astype(np.uint8)).save(os.path.join(GOLD, "lego_probe_100.png"))

    # ---- per-stage golden vectors on 64 rays spread over the probe image ---------------------
    sel = torch.linspace(0, 100 * 100 - 1, 64).long()
    rb64 = rb[sel].contiguous()
    g = {"ray_batch": rb64.numpy(), "c2w": c2w.numpy(), "focal": np.float64(focal)}
    for tag, o in (("det", O.RenderOpts(64, 128)),
                   ("rnd", O.RenderOpts(64, 128, perturb=True, radiance_field_noise_std=0.2, seed=42))):
        with torch.no_grad():
            r = O.render_rays(rb64, net_c, net_f, o)
        for k, v in r.items():
            g[f"{tag}_{k}"] = v.numpy()
    np.savez_compressed(os.path.join(GOLD, "lego_stage_vectors.npz"), **g)
    with open(os.path.join(GOLD, "lego_probe_stats.json"), "w") as f:
        json.dump(stats, f, indent=1)

    # ---- reference pose-parser fixtures (wheel dataset) -------------------------------------
    tmp = tempfile.mkdtemp()
    argv = sys.argv
    sys.argv = ["parser_instant_ngp.py", "--poses", f"{REF}/datasets/nerf-raw/wheel_real_robot/poses.txt",
                "--image_folder", f"{REF}/datasets/nerf-raw/wheel_real_robot/images_robot", "--recenter",
                "--scale_trans", "1.0", "--output", os.path.join(tmp, "transforms.json")]
    try:
        runpy.run_path(f"{REF}/parser_instant_ngp.py", run_name="__main__")
    finally:
        sys.argv = argv
    os.makedirs(os.path.join(GOLD, "wheel_transforms"), exist_ok=True)
    for fn in sorted(os.listdir(tmp)):
        with open(os.path.join(tmp, fn)) as f:
            txt = f.read().replace(f"{REF}/datasets/nerf-raw/wheel_real_robot/images_robot", "<IMAGES>")
        with open(os.path.join(GOLD, "wheel_transforms", fn), "w") as f:
            f.write(txt)
        print("wrote", fn)


if __name__ == "__main__":
    main()
