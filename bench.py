#!/usr/bin/env python3
"""Throughput of the render_rays training step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts the N ranks itself (one
fresh child process per GPU, before anything here touches a GPU) and rank 0 prints the JSON line.

Workload (the 400x400 lego configuration the throughput metric is quoted on): synthetic lego --
views of the reference's trained lego checkpoint rendered by the HIP path are the "photographs"
(the real nerf_synthetic images are not available offline) -- 64 coarse + 128 fine samples, coarse
and fine 8x256 skip networks with torch.nn.Linear default init, bf16 MFMA compute with fp32
accumulate and fp32 master weights.  One step = pick random pixels of one view, get_rays,
render_rays forward, img2mse on both maps, backward, (all-reduce of the flat gradient), Adam,
re-pack of the bf16 weight image: Trainer.step, i.e. nk_render_rays_fwd / nk_render_rays_bwd, the
path run_nerf.py trains with.  value = ray-samples/s = rays/s x (64+128), summed over all GPUs,
inputs resident in HBM.  --scaling weak (default): every GPU takes --rays rays per step;
--scaling strong: --rays is the global batch, split evenly over the GPUs.

Extra objects on the JSON line:
  roofline      the dominant kernel, the fused Embedder + MLP forward that saves activations
                (k_mlp_fwd<256,128,4,true>, two launches per step), priced against the bf16 MFMA
                peak: FLOP per launch / mean launch duration, from HIP events the library records
                around those launches inside the timed steps (nk_set_stage_events); the HBM view of
                the same launches (bytes written / duration) sits beside it under "hbm"
  kernels       mean duration of the other heavy launches of the timed steps, same events
  value_r1024   the same step at the reference's own batch size (num_random_rays: 1024), N == 1
  heldout_psnr  PSNR on held-out views after a short fixed-seed training run, N == 1
  cpu_baseline  the CPU restatement (oracle, torch CPU fp32) of the same training step, timed on
                this host on a bounded ray sample (rank 0, N == 1 only)
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_EVAL_8x256 = 1186816          # forward FLOPs of one 8x256 NeRF evaluation (2 x 593 408 MAC)
PEAK_BF16_TFLOPS = 2500.0              # MI355X dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0                  # MI355X HBM3E peak
MEASURED_HBM_WRITE_GBS = 5900.0        # streaming-write ceiling measured on this pool (tools/membench.hip)
ACT_BYTES_PER_POINT = 5056 + 320 + 16  # bytes one 8x256 evaluation writes in training mode
PMC_BYTES_PER_POINT = 5401.0           # measured (WRITE_SIZE + 2 FETCH_SIZE) / points, profiles/*pmc_hbm_bytes*.csv
EV = dict(fwd_c0=0, fwd_c1=1, fwd_f0=2, fwd_f1=3, bwd_fchain0=4, bwd_fchain1=5, bwd_fgemm1=6, bwd_cchain0=7,
          bwd_cchain1=8, bwd_end=9)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=16384, help="rays per step per GPU (global batch with --scaling strong)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--views", type=int, default=20, help="teacher views rendered for the synthetic dataset")
    ap.add_argument("--res", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=2048, help="rays in the CPU baseline sample")
    ap.add_argument("--no-extras", action="store_true", help="skip value_r1024 and heldout_psnr")
    ap.add_argument("--psnr-iters", type=int, default=2000)
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous, barrier and the one JSON line only, no GPU work (CPU test of the N-rank launch)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# N ranks from one command
# ------------------------------------------------------------------------------------------------
def launch_ranks(a) -> int:
    """Start a.gpus child processes of this script, one per GPU, and wait for them.  The parent never
    initialises a GPU; a child that fails takes the others down."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:           # exact PIDs this function started
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ------------------------------------------------------------------------------------------------
# legs that run on rank 0 of a 1-GPU job
# ------------------------------------------------------------------------------------------------
def cpu_baseline(O, rays, seed=0):
    """One training step of the CPU restatement on `rays` rays (8x256 pair, 64+128), median of 3."""
    import numpy as np
    import torch
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


def timed_steps(tr, sampler, steps, warmup, first=0):
    import torch
    for i in range(warmup):
        tr.step(*sampler.next(first + i))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(*sampler.next(first + warmup + i))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def heldout_psnr(nk, scene, dev, iters, rays, seed):
    """Train a fresh 8x256 pair for `iters` iterations on the scene's training views and return the
    mean PSNR of the fine map on its held-out views (validation settings: no jitter, no noise)."""
    import torch
    from nerf_kinematics_amd.datasets import RaySampler
    from nerf_kinematics_amd.metrics import mse2psnr
    from nerf_kinematics_amd.trainer import Trainer, TrainOpts
    opts = TrainOpts(num_coarse=64, num_fine=128, num_random_rays=rays, perturb=True, radiance_field_noise_std=0.2,
                     lr=5e-4, seed=seed)
    tr = Trainer(nk.ARCH_RUN_NERF, nk.ARCH_RUN_NERF, opts, dev, rays_per_step=rays)
    smp = RaySampler(scene, rays, seed=seed)
    t0 = time.perf_counter()
    for i in range(iters):
        tr.step(*smp.next(i))
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    vals = []
    for k in scene.i_val:
        out = tr.render(scene.rays(int(k), dev))
        vals.append(mse2psnr(float(((out["rgb_fine"] - scene.target(int(k))) ** 2).mean())))
    return sum(vals) / len(vals), train_s


# ------------------------------------------------------------------------------------------------
def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    if os.environ.get("NK_STALL_DUMP_SECONDS"):     # a rank still running after n seconds dumps its stacks and exits
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["NK_STALL_DUMP_SECONDS"]), exit=True)

    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("NK_DIST_BACKEND") or ("gloo" if a.launch_check else "nccl")
    if a.launch_check:
        # the launch path alone: rendezvous, barrier, max-over-ranks, one line from rank 0
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)
            torch.distributed.barrier()
            t = torch.tensor([float(rank)], dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            assert int(t.item()) == world - 1
        if rank == 0:
            print(json.dumps({"metric": "ray-samples/sec (train)", "value": 0.0, "unit": "ray-samples/s", "n_gpus": world,
                              "steps": 0, "warmup": 0, "launch_check": True, "scaling": a.scaling}), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (there is no CPU path to measure)")
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks over RCCL need {world} GPUs, this node shows {ndev} "
                         "(NK_DIST_BACKEND=gloo lets ranks share a GPU, for tests)")
    dev = torch.device(f"cuda:{local % ndev}")
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": dev} if backend == "nccl" else {}
        torch.distributed.init_process_group(backend, rank=rank, world_size=world, **kw)

    import nerf_kinematics_amd as nk
    from nerf_kinematics_amd import lib as nklib
    from nerf_kinematics_amd.datasets import RaySampler, teacher_scene
    from nerf_kinematics_amd.trainer import Trainer, TrainOpts
    L = nk.load()

    rays = a.rays if a.scaling == "weak" else max(a.rays // world, 1)
    # the synthetic photographs: the trained lego checkpoint seen from a ring of poses (8-bit, in HBM)
    scene = teacher_scene(os.path.join(ROOT, "tests", "golden", "lego_lowres3_weights.npz"), a.views, 2, a.res, dev,
                          phase=0.37 * rank)
    opts = TrainOpts(num_coarse=64, num_fine=128, num_random_rays=rays, perturb=True,
                     radiance_field_noise_std=0.2, lr=5e-4, seed=42)
    tr = Trainer(nk.ARCH_RUN_NERF, nk.ARCH_RUN_NERF, opts, dev, rays_per_step=rays)
    sampler = RaySampler(scene, rays, seed=1234, rank=rank, world=world)

    # HIP events the library records at the stage boundaries of nk_render_rays_fwd / _bwd: torch
    # creates them (a first record() materialises the handle), the library re-records them on the
    # stream the kernels run on
    import ctypes
    n_ev = len(EV)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(n_ev)] for _ in range(a.steps)]
    for row in ev:
        for e in row:
            e.record()
    torch.cuda.synchronize()
    handles = [(ctypes.c_void_p * n_ev)(*[e.cuda_event for e in row]) for row in ev]

    for i in range(a.warmup):
        tr.step(*sampler.next(i))
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        nklib.check(L.nk_set_stage_events(handles[i], n_ev))
        loss = tr.step(*sampler.next(a.warmup + i))
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    nklib.check(L.nk_set_stage_events(None, 0))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        samples = rays * 192 * a.steps * world
        value = samples / dt
        span = lambda x, y: float(np.mean([row[EV[x]].elapsed_time(row[EV[y]]) for row in ev]))
        t_f, t_c = span("fwd_f0", "fwd_f1"), span("fwd_c0", "fwd_c1")
        # Roofline kernel: k_mlp_fwd<256,128,4,true>, two launches per step (coarse: 64 evaluations
        # per ray, fine: 192).  Per point it does 1 186 816 FLOP and writes 5 392 B (5 056 B of bf16
        # activations + 320 B of ReLU bit words + 16 B of raw output; rays, depths and the L2-resident
        # weight stream are noise).  Launch duration = mean over both launches of every timed step.
        k_ms = 0.5 * (t_f + t_c)
        pts = rays * (64 + 192) / 2.0                           # points per launch, averaged like k_ms
        flops = pts * FLOP_PER_EVAL_8x256
        tflops = flops / (k_ms * 1e-3) / 1e12
        bytes_per_launch = pts * ACT_BYTES_PER_POINT
        gbs = bytes_per_launch / (k_ms * 1e-3) / 1e9
        shared = rays * 192 <= (1 << 20)                        # weight-gradient products of both networks in one shared launch
        res = {
            "metric": "ray-samples/sec (train)", "value": value, "unit": "ray-samples/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": a.scaling, "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"lego-teacher {a.res}x{a.res}, 64 coarse + 128 fine, 8x256 coarse+fine NeRF, "
                                   f"train step (fwd+bwd+Adam), {rays} rays/step/GPU",
                       "rays_per_step_per_gpu": rays, "num_coarse": 64, "num_fine": 128,
                       "mlp_evals_per_s": value / 192 * 256, "final_loss": float(loss),
                       "step_path": "Trainer.step -> nk_render_rays_fwd / nk_render_rays_bwd"},
            "roofline": {"bound": "mfma", "kernel": "k_mlp_fwd<256,128,4,true> (Embedder + MLP forward, activations saved)",
                         "achieved": tflops, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": tflops / PEAK_BF16_TFLOPS,
                         # HBM bytes per launch from the PMC passes kept under profiles/ (WRITE_SIZE +
                         # 2 x FETCH_SIZE of the fine launch, scaled to this mean launch)
                         "traffic": pts * PMC_BYTES_PER_POINT,
                         "kernel_ms": k_ms, "kernel_ms_fine": t_f, "kernel_ms_coarse": t_c, "flop_per_launch": flops,
                         "hbm": {"achieved": gbs, "unit": "GB/s", "peak": PEAK_HBM_GBS, "frac": gbs / PEAK_HBM_GBS,
                                 "measured_write_ceiling": MEASURED_HBM_WRITE_GBS,
                                 "frac_of_measured": gbs / MEASURED_HBM_WRITE_GBS, "bytes_per_launch": bytes_per_launch},
                         "step_train_tflops": value * 3 * FLOP_PER_EVAL_8x256 * 256 / 192 / 1e12,
                         "step_frac_of_mfma_peak": value * 3 * FLOP_PER_EVAL_8x256 * 256 / 192 / 1e12 / PEAK_BF16_TFLOPS},
            # the other heavy launches of the same timed steps (ms, mean): backward chains (4 928 + 180 B
            # per point) and weight-gradient products + reduce (10 688 B read per point)
            "kernels": {"bwd_chain_fine_ms": span("bwd_fchain0", "bwd_fchain1"),
                        "bwd_chain_coarse_ms": span("bwd_cchain0", "bwd_cchain1"),
                        "dw_gemms_fine_ms": None if shared else span("bwd_fchain1", "bwd_fgemm1"),
                        "dw_gemms_coarse_ms": None if shared else span("bwd_cchain1", "bwd_end"),
                        "dw_gemms_shared_ms": span("bwd_cchain1", "bwd_end") if shared else None,
                        "bwd_chain_fine_gbs": rays * 192 * (4928 + 180) / (span("bwd_fchain0", "bwd_fchain1") * 1e-3) / 1e9},
        }
        if world == 1 and not a.no_extras:
            tr = None
            torch.cuda.empty_cache()
            opts1k = TrainOpts(num_coarse=64, num_fine=128, num_random_rays=1024, perturb=True,
                               radiance_field_noise_std=0.2, lr=5e-4, seed=42)
            tr1k = Trainer(nk.ARCH_RUN_NERF, nk.ARCH_RUN_NERF, opts1k, dev, rays_per_step=1024)
            t1k = timed_steps(tr1k, RaySampler(scene, 1024, seed=1234), 300, 30)
            res["value_r1024"] = {"value": 1024 * 192 / t1k, "unit": "ray-samples/s", "ms_per_step": t1k * 1e3,
                                  "rays_per_step": 1024, "steps": 300, "warmup": 30}
            tr1k = None
            p, train_s = heldout_psnr(nk, scene, dev, a.psnr_iters, 4096, seed=42)
            res["heldout_psnr"] = {"value": p, "unit": "dB", "iters": a.psnr_iters, "rays_per_step": 4096, "seed": 42,
                                   "train_views": a.views, "heldout_views": 2, "train_seconds": train_s,
                                   "note": "fresh 8x256 pair on the lego-teacher views, fine map, validation settings"}
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
