#!/usr/bin/env python3
"""Throughput of the render_rays training step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N == 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (the 400x400 lego configuration the throughput metric is quoted on): synthetic lego --
views of the reference's trained lego checkpoint rendered by the HIP path are the "photographs" (the real nerf_synthetic images are
not available offline) -- 64 coarse + 128 fine samples, coarse and fine 8x256 skip networks with
torch.nn.Linear default init, bf16 MFMA compute with fp32 accumulate and fp32 master weights.
One step = pick random pixels of one view, get_rays, render_rays forward, img2mse on both maps,
backward, (all-reduce of the flat gradient), Adam, re-pack of the bf16 weight image.
Every GPU takes --rays rays per step (weak scaling).  value = ray-samples/s = rays/s x (64+128),
summed over all GPUs, inputs resident in HBM.

Extra objects on the JSON line:
  roofline      the fused MLP forward launches that save activations (k_mlp_fwd<..,true>, two per
                step): algorithmic bytes per launch / mean launch duration from HIP events recorded
                around them in the timed steps, against the HBM peak (the roofline that bounds it)
  cpu_baseline  the CPU restatement (oracle, torch CPU fp32) of the same training step, timed on
                this host on a bounded ray sample (rank 0, N == 1 only)
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_EVAL_8x256 = 1186816          # forward FLOPs of one 8x256 NeRF evaluation (2 x 593 408 MAC)
PEAK_BF16_TFLOPS = 2500.0              # MI355X dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0                  # MI355X HBM3E peak
ACT_BYTES_PER_POINT = 5056 + 320 + 16  # bytes one 8x256 evaluation writes in training mode (see roofline below)
PMC_BYTES_PER_POINT = 5401.0           # measured: (WRITE_SIZE + 2 FETCH_SIZE) / points, profiles/r01_pmc_hbm_bytes.csv


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=16384, help="rays per step per GPU")
    ap.add_argument("--views", type=int, default=4, help="teacher views rendered for the synthetic dataset")
    ap.add_argument("--res", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=2048, help="rays in the CPU baseline sample")
    return ap.parse_args()


def cpu_baseline(O, rays, seed=0):
    """One training step of the CPU restatement on `rays` rays (8x256 pair, 64+128), median of 3."""
    torch.manual_seed(seed)
    arch = O.ARCH_RUN_NERF
    fc = O.NeRFNet.init_flat(arch, 42).requires_grad_(True)
    ff = O.NeRFNet.init_flat(arch, 43).requires_grad_(True)
    focal = 0.5 * 400 / math.tan(0.5 * 0.6911112070083618)
    ro, rd = O.get_rays(400, 400, focal, O.spherical_pose(45.0, -30.0, 4.0))
    rb = O.make_ray_batch(ro, rd, 2.0, 6.0)
    sel = torch.randperm(rb.shape[0])[:rays]
    rb = rb[sel].contiguous()
    target = torch.rand(rays, 3)
    opt = torch.optim.Adam([fc, ff], lr=5e-4)
    times = []
    for it in range(4):
        t0 = time.perf_counter()
        co, fo = O.NeRFNet.from_flat(arch, fc), O.NeRFNet.from_flat(arch, ff)
        out = O.render_rays(rb, co, fo, O.RenderOpts(64, 128, perturb=True, radiance_field_noise_std=0.2, seed=it))
        loss = O.nerf_loss(out, target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[1:]))
    return rays * 192 / t, t


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    import nerf_kinematics_amd as nk
    from nerf_kinematics_amd.trainer import Trainer, TrainOpts

    from nerf_kinematics_amd.datasets import RaySampler, teacher_scene
    # the synthetic photographs: the trained lego checkpoint seen from a ring of poses (8-bit, in HBM)
    scene = teacher_scene(os.path.join(ROOT, "tests", "golden", "lego_lowres3_weights.npz"), a.views, 0, a.res, dev,
                          phase=0.37 * rank)
    opts = TrainOpts(num_coarse=64, num_fine=128, num_random_rays=a.rays, perturb=True,
                     radiance_field_noise_std=0.2, lr=5e-4, seed=42)
    tr = Trainer(nk.ARCH_RUN_NERF, nk.ARCH_RUN_NERF, opts, dev, rays_per_step=a.rays)
    sampler = RaySampler(scene, a.rays, seed=1234, rank=rank, world=world)

    mk_ev = lambda: tuple(torch.cuda.Event(enable_timing=True) for _ in range(6))
    ev = [mk_ev() for _ in range(a.steps)]       # (fine fwd start/stop, coarse fwd start/stop, fine bwd start/stop) per timed step
    dummy = mk_ev()

    def one_step(i, events):
        rb, target = sampler.next(i)        # one kernel: pixel draw + get_rays + target colours
        return tr.step(rb, target, events=events)

    for i in range(a.warmup):
        one_step(i, dummy)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = one_step(a.warmup + i, ev[i])
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        samples = a.rays * 192 * a.steps * world
        value = samples / dt
        # Roofline kernel: k_mlp_fwd<256,128,4,true>, the fused Embedder + MLP forward that saves bf16
        # activations for backward; two launches per step (coarse: 64 evaluations per ray, fine: 192).
        # Per point it does 1 186 816 FLOP and moves 5 392 B (5 056 B of bf16 activations + 320 B of
        # ReLU bit words written, 16 B of raw output written; rays / depths / the L2-resident weight
        # stream are noise): 0.47 ns at the MFMA peak, 0.67 ns at the HBM peak -- HBM is the roofline
        # that bounds it.  The launch duration is the mean over both launches of every timed step
        # (HIP events on the launch stream), which is what a kernel trace averages too.
        t_f = [e[0].elapsed_time(e[1]) for e in ev]
        t_c = [e[2].elapsed_time(e[3]) for e in ev]
        k_ms = float(np.mean(t_f + t_c))
        pts = a.rays * (64 + 192) / 2.0                         # points per launch, averaged like k_ms
        bytes_per_launch = pts * ACT_BYTES_PER_POINT
        achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9
        flops = pts * FLOP_PER_EVAL_8x256
        res = {
            "metric": "ray-samples/sec (train)", "value": value, "unit": "ray-samples/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"lego-teacher {a.res}x{a.res}, 64 coarse + 128 fine, 8x256 coarse+fine NeRF, "
                                   f"train step (fwd+bwd+Adam), {a.rays} rays/step/GPU",
                       "rays_per_step_per_gpu": a.rays, "num_coarse": 64, "num_fine": 128,
                       "mlp_evals_per_s": value / 192 * 256, "final_loss": float(loss)},
            "roofline": {"bound": "hbm", "kernel": "k_mlp_fwd<256,128,4,true> (Embedder + MLP forward, activations saved)",
                         "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                         # HBM bytes per launch from the PMC passes kept under profiles/ (WRITE_SIZE +
                         # 2 x FETCH_SIZE of the fine launch, scaled to this mean launch): all of it algorithmic
                         "traffic": pts * PMC_BYTES_PER_POINT,
                         "kernel_ms": k_ms, "kernel_ms_fine": float(np.mean(t_f)), "kernel_ms_coarse": float(np.mean(t_c)),
                         "bytes_per_launch": bytes_per_launch, "flop_per_launch": flops,
                         "mfma_tflops": flops / (k_ms * 1e-3) / 1e12, "mfma_frac": flops / (k_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                         "step_train_tflops": samples / dt * 3 * FLOP_PER_EVAL_8x256 * 256 / 192 / 1e12,
                         # the rest of the step, for orientation: the fine network's MLP backward (dZ chain +
                         # weight-gradient GEMMs) moves 4 928 + 180 B (chain) and 10 688 B (GEMM reads) per point
                         "backward_fine_ms": float(np.mean([e[4].elapsed_time(e[5]) for e in ev])),
                         "backward_fine_gbs": a.rays * 192 * (4928 + 180 + 10688) / (float(np.mean([e[4].elapsed_time(e[5]) for e in ev])) * 1e-3) / 1e9},
        }
        if world == 1 and not a.no_cpu_baseline:
            torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))))
            from oracle import nerf_oracle as O        # the CPU restatement is only ever the baseline leg
            v, t = cpu_baseline(O, a.cpu_rays)
            res["cpu_baseline"] = {"value": v, "unit": "ray-samples/s", "cores": torch.get_num_threads(),
                                   "kind": "port",
                                   "sample": f"{a.cpu_rays} rays x (64+128) samples, one training step "
                                             f"(fwd+bwd+Adam) of the torch-CPU fp32 restatement, median of 3 "
                                             f"after 1 warm-up, {t:.2f} s/step"}
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
