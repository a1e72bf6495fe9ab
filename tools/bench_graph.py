"""Eager Trainer.step against the graph-replayed step (dev tool).
    python tools/bench_graph.py <config.yml> [--rays N] [--as-trained]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_kinematics_amd import config as C
from nerf_kinematics_amd.datasets import RaySampler, teacher_scene
from nerf_kinematics_amd.model import Arch
from nerf_kinematics_amd.trainer import GraphedStep, Trainer, TrainOpts
ap = argparse.ArgumentParser()
ap.add_argument("config"); ap.add_argument("--rays", type=int, default=None); ap.add_argument("--steps", type=int, default=500)
ap.add_argument("--as-trained", action="store_true")
a = ap.parse_args()
cfg = C.load_config(a.config); opts = TrainOpts.from_config(cfg)
if a.rays: opts.num_random_rays = a.rays
dev = torch.device("cuda:0")
mc = cfg["models"]["coarse"]; arch = Arch.from_config(mc)
if not a.as_trained and mc["type"] == "FlexibleNeRFModel":
    arch = Arch(int(mc["num_layers"]), arch.width, -1, arch.dir_width, arch.n_freq_xyz, arch.n_freq_dir, False, True, "flexible")
scene = teacher_scene(None, 8, 0, 200, dev)
n = opts.num_coarse + opts.num_fine
for mode in ("eager", "graph"):
    tr = Trainer(arch, arch, opts, dev); smp = RaySampler(scene, opts.num_random_rays, opts.seed)
    gs = GraphedStep(tr, smp) if mode == "graph" else None
    step = (lambda i: gs.step()) if gs else (lambda i: tr.step(*smp.next(i)))
    for i in range(30): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(a.steps): step(30 + i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    print(f"{mode:6s} depth {arch.depth} width {arch.width} {opts.num_coarse}+{opts.num_fine} {opts.num_random_rays} rays: "
          f"{dt * 1e3:.3f} ms/step, {opts.num_random_rays * n / dt / 1e6:.1f} M ray-samples/s, loss {float(tr.loss):.5f}")
