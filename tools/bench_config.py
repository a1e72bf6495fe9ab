"""Training-step throughput for a reference experiment file (dev tool): the networks, sample counts
and rays per step of the YAML, on the synthetic teacher scene.
    python tools/bench_config.py tests/golden/lego-lowres3_config.yml [--rays N] [--steps K]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_kinematics_amd import lib as _L
if os.environ.get("NK_LIB"):            # experiment builds
    _L.LIB_PATH = os.path.abspath(os.environ["NK_LIB"])
from nerf_kinematics_amd import config as C
from nerf_kinematics_amd.datasets import RaySampler, teacher_scene
from nerf_kinematics_amd.model import Arch, ARCH_LEGO_CKPT, ARCH_WHEEL_CKPT
from nerf_kinematics_amd.trainer import Trainer, TrainOpts

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("--rays", type=int, default=None)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--as-trained", action="store_true", help="4 trunk layers as in the reference's checkpoints (default: as the YAML says)")
ap.add_argument("--scene", action="store_true", help="train on the dataset the YAML names (default: a small lego-teacher scene)")
ap.add_argument("--downsample", type=int, default=None, help="override dataset.downsample_factor")
a = ap.parse_args()
cfg = C.load_config(a.config)
opts = TrainOpts.from_config(cfg)
if a.rays:
    opts.num_random_rays = a.rays
dev = torch.device("cuda:0")
mc = cfg["models"]["coarse"]
arch = Arch.from_config(mc)
if not a.as_trained and mc["type"] == "FlexibleNeRFModel":
    arch = Arch(int(mc["num_layers"]), arch.width, -1, arch.dir_width, arch.n_freq_xyz, arch.n_freq_dir, False, True, "flexible")
tr = Trainer(arch, arch, opts, dev)
if a.scene:
    from nerf_kinematics_amd.datasets import load_scene
    if a.downsample is not None:
        cfg["dataset"]["downsample_factor"] = a.downsample
    scene = load_scene(cfg, dev)
    print(f"scene: {len(scene.i_train)} training views {scene.W}x{scene.H}, ndc={scene.ndc}")
else:
    scene = teacher_scene(None, 8, 0, 200, dev)
smp = RaySampler(scene, opts.num_random_rays, opts.seed)
for i in range(20):
    tr.step(*smp.next(i))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    tr.step(*smp.next(20 + i))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
n = opts.num_coarse + opts.num_fine
print(f"{os.path.basename(a.config)}: depth {arch.depth} width {arch.width}, {opts.num_coarse}+{opts.num_fine} samples, "
      f"{opts.num_random_rays} rays/step: {dt * 1e3:.3f} ms/step, {1 / dt:.0f} it/s, {opts.num_random_rays * n / dt / 1e6:.1f} M ray-samples/s")
