#!/usr/bin/env python3
"""Train / render entry point: ``python run_nerf.py --config configs/lego.yml``.

Same invocation as the reference's only documented call site (/root/reference/README.md:90-93)
and driven by the same experiment file
(/root/reference/trained_models/nerf-pytorch/lego-lowres3/config.yml): ``experiment.train_iters``,
``print_every``, ``validate_every``, ``save_every``, ``logdir`` / ``id`` (:10-17); the run writes
``<logdir>/<id>/config.yml`` and ``checkpointNNNNN.ckpt`` files in the reference's layout.
Scalars go to ``<logdir>/<id>/log.jsonl`` (one JSON object per line) instead of TensorBoard event
files.

One process per GPU: under ``torch.distributed.run`` every rank draws its own rays and the flat
gradient is all-reduced (RCCL); whole-image renders are sharded by rows and all-gathered; rank 0
logs and saves.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from typing import Optional

import numpy as np
import torch

from . import checkpoint as ckpt
from . import config as config_mod
from . import dist as nkdist
from .datasets import RaySampler, Scene, load_scene, read_transforms
from .metrics import mse2psnr, psnr_8bit, to_8bit
from .model import Arch
from .trainer import Trainer, TrainOpts


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="NeRF training / rendering on MI355X")
    ap.add_argument("--config", required=True, help="experiment file (reference YAML schema)")
    ap.add_argument("--load-checkpoint", "--load_checkpoint", default="", help=".ckpt to resume from / render")
    ap.add_argument("--render-only", action="store_true", help="no training: render --render-poses or the test split")
    ap.add_argument("--render-poses", default="", help="transforms json whose frames are rendered (test / video paths)")
    ap.add_argument("--iters", type=int, default=None, help="override experiment.train_iters")
    ap.add_argument("--rays", type=int, default=None, help="override nerf.train.num_random_rays (per GPU)")
    ap.add_argument("--logdir", default=None, help="override experiment.logdir")
    ap.add_argument("--device", default=None, help="cuda:N (default: LOCAL_RANK)")
    ap.add_argument("--act-format", choices=config_mod.ACT_FORMATS, default=None,
                    help="override nerf.train.act_format: format of the tensors saved for backward "
                         "(bf16 = default; fp8 / fp8bf8 = experimental 8-bit formats)")
    ap.add_argument("--skip-empty-eps", type=float, default=None,
                    help="override nerf.validation.skip_empty_eps: renders skip the fine network where every "
                         "nearby coarse weight is below this value (0 = off, the default; experiment)")
    ap.add_argument("--graph", action="store_true",
                    help="single GPU: replay the training iteration as one HIP graph (same results bit for bit)")
    return ap.parse_args(argv)


def build_archs(cfg: dict, ck: Optional[dict]):
    mc, mf = cfg["models"]["coarse"], cfg["models"].get("fine")
    if ck is not None:
        ac = ckpt.arch_for_state_dict(mc, ck["model_coarse_state_dict"])
        af = ckpt.arch_for_state_dict(mf, ck["model_fine_state_dict"]) if (mf and ck.get("model_fine_state_dict")) else None
    else:
        ac = Arch.from_config(mc)
        af = Arch.from_config(mf) if mf else None
    return ac, af


def save_png(path: str, rgb8: np.ndarray) -> None:
    from PIL import Image
    Image.fromarray(rgb8).save(path)


@torch.no_grad()
def render_view(tr: Trainer, scene: Scene, cfg: dict, k: Optional[int] = None, pose=None) -> torch.Tensor:
    """Whole image [H, W, 3] float of view k (or an explicit pose) with the validation settings."""
    kw = config_mod.render_kwargs(cfg, "validation")
    chunk = int(cfg["nerf"]["validation"].get("chunksize", 131072))
    rb = scene.rays(0 if k is None else k, tr.device, pose=pose)
    out = tr.render(rb, chunk=max(1024, min(chunk, 65536)), opts=kw, shard=True)    # every rank takes a block of rows
    return out["rgb_fine"].view(scene.H, scene.W, 3)


def validate(tr: Trainer, scene: Scene, cfg: dict, k: int):
    img = render_view(tr, scene, cfg, k)
    target = scene.target(k).view(scene.H, scene.W, 3)
    mse = float(((img - target) ** 2).mean())
    p8 = psnr_8bit(to_8bit(target.cpu().numpy()), to_8bit(img.cpu().numpy()))
    return mse, mse2psnr(mse), p8, img


def render_path(tr: Trainer, scene: Scene, cfg: dict, poses_json: str, outdir: str, write: bool = True) -> int:
    """Render every frame of a transforms json (keys ``transform_matrix`` or
    ``transform_matrix_start``, as the pose converter writes for its test / video rings).  Called
    by every rank (each renders its rows of every frame); ``write`` is true on the rank that saves."""
    if write:
        os.makedirs(outdir, exist_ok=True)
    frames = read_transforms(poses_json)["frames"]
    n = 0
    for i, fr in enumerate(frames):
        m = fr.get("transform_matrix", fr.get("transform_matrix_start"))
        if m is None:
            continue
        img = render_view(tr, scene, cfg, pose=np.array(m, dtype=np.float32))
        if write:
            save_png(os.path.join(outdir, f"{i:04d}.png"), to_8bit(img.cpu().numpy()))
        n += 1
    return n


def main(argv=None) -> int:
    a = parse_args(argv)
    # NK_STALL_DUMP_SECONDS=n: if the process is still running after n seconds, every thread's
    # Python stack goes to stderr and the process exits (multi-rank runs: shows where each rank waits)
    if os.environ.get("NK_STALL_DUMP_SECONDS"):
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["NK_STALL_DUMP_SECONDS"]), exit=True)
    over = {}
    if a.logdir:
        over.setdefault("experiment", {})["logdir"] = a.logdir
    if a.iters is not None:
        over.setdefault("experiment", {})["train_iters"] = a.iters
    if a.rays is not None:
        over.setdefault("nerf", {}).setdefault("train", {})["num_random_rays"] = a.rays
    if a.act_format is not None:
        over.setdefault("nerf", {}).setdefault("train", {})["act_format"] = a.act_format
    if a.skip_empty_eps is not None:
        over.setdefault("nerf", {}).setdefault("validation", {})["skip_empty_eps"] = a.skip_empty_eps
    cfg = config_mod.load_config(a.config, over)

    rank, world, local = nkdist.init_from_env()
    dev = torch.device(a.device or f"cuda:{local}")
    torch.cuda.set_device(dev)
    exp = cfg["experiment"]
    logdir = os.path.join(exp["logdir"], str(exp["id"]))
    if rank == 0:
        os.makedirs(logdir, exist_ok=True)
        config_mod.dump_config(cfg, os.path.join(logdir, "config.yml"))

    ck = ckpt.load_checkpoint(a.load_checkpoint) if a.load_checkpoint else None
    arch_c, arch_f = build_archs(cfg, ck)
    opts = TrainOpts.from_config(cfg)
    tr = Trainer(arch_c, arch_f, opts, dev)
    if ck is not None:
        tr.import_state(ckpt.state_from_checkpoint(ck, arch_c, arch_f))
        if rank == 0:
            print(f"resumed {a.load_checkpoint} at iteration {tr.iteration}", flush=True)
    scene = load_scene(cfg, dev)

    if a.render_only:           # every rank renders its block of rows of every frame; rank 0 writes
        out = os.path.join(logdir, "render")
        if a.render_poses:
            n = render_path(tr, scene, cfg, a.render_poses, out, write=rank == 0)
        else:
            if rank == 0:
                os.makedirs(out, exist_ok=True)
            n = 0
            for k in scene.i_test:
                mse, psnr, p8, img = validate(tr, scene, cfg, int(k))
                if rank == 0:
                    save_png(os.path.join(out, f"{int(k):04d}.png"), to_8bit(img.cpu().numpy()))
                    print(f"view {int(k)}: mse {mse:.6f} psnr {psnr:.3f} dB psnr8 {p8:.3f} dB", flush=True)
                n += 1
        if rank == 0:
            print(f"rendered {n} frames into {out}", flush=True)
        nkdist.shutdown()
        return 0

    sampler = RaySampler(scene, opts.num_random_rays, opts.seed, rank, world)
    graphed = None
    if a.graph:
        if world > 1:
            raise SystemExit("--graph: single-process training only")
        from .trainer import GraphedStep
        graphed = GraphedStep(tr, sampler)
    log = open(os.path.join(logdir, "log.jsonl"), "a") if rank == 0 else None
    if world > 1:
        tr.comm_events = []         # events around the gradient all-reduce until the first report
    iters = int(exp["train_iters"])
    t_last, it_last = time.perf_counter(), tr.iteration
    loss_val = float("nan")
    for i in range(tr.iteration, iters):
        if graphed is not None:
            loss = graphed.step()
        else:
            rb, target = sampler.next(i)
            loss = tr.step(rb, target)
        last = i == iters - 1
        if rank == 0 and (i % int(exp["print_every"]) == 0 or last):
            loss_val = float(loss)            # the only host sync of the loop
            now = time.perf_counter()
            rate = (i + 1 - it_last) * opts.num_random_rays * world / max(now - t_last, 1e-9)
            t_last, it_last = now, i + 1
            # the loss is the sum of the coarse and the fine map's MSE; PSNR is quoted on that scalar,
            # like the ``psnr`` stored in the reference's checkpoints
            rec = {"iter": i, "loss": loss_val, "psnr": mse2psnr(loss_val), "rays_per_s": rate,
                   "ray_samples_per_s": rate * (opts.num_coarse + opts.num_fine)}
            log.write(json.dumps({"train": rec}) + "\n"); log.flush()
            print(f"[TRAIN] Iter: {i} Loss: {loss_val:.6f} PSNR: {rec['psnr']:.3f} rays/s: {rate:.0f}", flush=True)
            if tr.comm_events and len(tr.comm_events) > 1:      # once: where a multi-rank step's time goes (the loss read above has synchronised)
                ar = [sum(b.elapsed_time(e) for b, e in pairs) for pairs in tr.comm_events[1:]] or [0.0]
                try:
                    ver = ".".join(str(v) for v in torch.cuda.nccl.version())
                except Exception:        # noqa: BLE001
                    ver = "unknown"
                comm = {"world": world, "backend": torch.distributed.get_backend(), "rccl": ver,
                        "allreduce_ms": sum(ar) / len(ar), "steps": len(ar), "gradient_bytes": int(tr.grad.numel()) * 4,
                        "split_allreduce": bool(tr._split_allreduce(tr.R)),
                        "NCCL_ALGO": os.environ.get("NCCL_ALGO"), "NCCL_PROTO": os.environ.get("NCCL_PROTO")}
                log.write(json.dumps({"comm": comm}) + "\n"); log.flush()
                print(f"[COMM] {json.dumps(comm)}", flush=True)
                tr.comm_events = None
        if tr.comm_events is not None and len(tr.comm_events) > 256:
            if rank == 0:
                del tr.comm_events[256:]        # keep the first steps until the next report
            else:
                tr.comm_events = None           # only rank 0 reports
        if len(scene.i_val) and ((i % int(exp["validate_every"]) == 0 and i > 0) or last):
            k = int(scene.i_val[(i // max(int(exp["validate_every"]), 1)) % len(scene.i_val)])
            mse, psnr, p8, img = validate(tr, scene, cfg, k)        # all ranks: the render is sharded by rows
            if rank == 0:
                log.write(json.dumps({"validation": {"iter": i, "view": k, "loss": mse, "psnr": psnr, "psnr_8bit": p8}}) + "\n")
                log.flush()
                print(f"[VAL] Iter: {i} view {k} Loss: {mse:.6f} PSNR: {psnr:.3f} PSNR(8-bit): {p8:.3f}", flush=True)
        if rank == 0 and ((i % int(exp["save_every"]) == 0 and i > 0) or last):
            lv = float(loss)
            path = os.path.join(logdir, ckpt.checkpoint_name(i))
            ckpt.save_checkpoint(path, tr.export_state(lv, mse2psnr(lv)), arch_c, arch_f, betas=opts.betas, eps=opts.eps)
            print(f"saved {path}", flush=True)
    if log:
        log.close()
    nkdist.shutdown()
    return 0


if __name__ == "__main__":
    sys.exit(main())
