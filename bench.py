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
  roofline      the time-dominant kernel of the step, k_dw_gemm (all weight-gradient products of a
                network in one launch + its k_dw_reduce; two launches per step), priced against the
                bf16 MFMA peak: FLOP per launch / mean launch duration, from HIP events the library
                records at the stage boundaries inside the timed steps (nk_set_stage_events); its HBM
                view (bytes read / duration) sits beside it under "hbm", and "per_kernel" holds the
                same two views of the three heavy launches (forward chain, backward chain, dW)
  value_long    the same step timed over >= 200 further steps (the driver's --steps 20 is 0.4 s)
  value_8bit_formats  the same step with the experimental 8-bit saved-tensor formats (never `value`), N == 1
  value_r1024   the same step at the reference's own batch size (num_random_rays: 1024), N == 1
  render_400x400_ms  whole-image inference (nerf.validation settings), one 400x400 view, N == 1
  heldout_psnr  PSNR on held-out views after a short fixed-seed training run, N == 1
  cpu_baseline  the CPU restatement (oracle, torch CPU fp32) of the same training step, timed on
                this host on a bounded ray sample (rank 0, N == 1 only)
  comm          N > 1 only: all-reduce time per step, per-rank step times, RCCL version and settings
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

FLOP_PER_EVAL_8x256 = 1186816          # forward FLOPs of one 8x256 NeRF evaluation (2 x 593 408 MAC); the
                                       # weight-gradient products dW = dZ^T X of one evaluation cost the same
FLOP_BWD_CHAIN_8x256 = 1115392         # dH = W^T dZ through rgb, views (feature part), feature + alpha, trunk 7..1
PEAK_BF16_TFLOPS = 2500.0              # MI355X dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0                  # MI355X HBM3E peak
MEASURED_HBM_WRITE_GBS = 5900.0        # streaming-write ceiling measured on this pool (tools/membench.hip)
MEASURED_HBM_READ_GBS = 6200.0         # streaming-read ceiling, same tool
# HBM bytes per point-evaluation of an 8x256 network, by saved-tensor format (see ACT_FORMATS):
#   forward writes the saved activations + 320 B of ReLU bit words + 16 B of raw output,
#   the backward chain writes dZ and reads the bit words and 16 B of d_raw,
#   the weight-gradient products read every dZ tensor and every layer input once
#   (gamma(x) twice: first layer and skip layer; h_7 and dOut shared by feature / alpha).
ACT_FEATS, DZ_FEATS = 2528, 2464               # features saved by the forward / written by the backward chain per point
GEMM_DZ_FEATS, GEMM_ACT_FEATS = 2496, 2592     # features the 11 products read per point (dZ side, layer-input side)
ACT_FORMATS = {"bf16": (2, 2), "fp8": (1, 2), "fp8bf8": (1, 1)}      # bytes per saved activation, per dZ element
# k_dw_gemm's measured HBM bytes per point ((WRITE_SIZE + 2 x FETCH_SIZE) / points of a launch), from
# separate rocprofv3 --pmc passes of this command; formats without a pass report traffic null
PMC_DW_BYTES_PER_POINT = {"bf16": 10206.0, "fp8bf8": 5118.0}
PMC_SOURCE = ("profiles/r03_kernel_table*.csv (separate rocprofv3 --pmc passes of this command, tools/profile_bench.sh), "
              "not counted in this run")
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
    ap.add_argument("--long-steps", type=int, default=200, help="steps of the value_long leg")
    ap.add_argument("--act-format", choices=tuple(ACT_FORMATS), default="bf16",
                    help="format of the tensors saved for backward: bf16 (default), fp8 = e4m3 activations, "
                         "fp8bf8 = e4m3 activations and e5m2 dZ (experiments, off by default)")
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


def heldout_psnr(nk, scene, dev, iters, rays, seed, act_format="bf16"):
    """Train a fresh 8x256 pair for `iters` iterations on the scene's training views and return the
    mean PSNR of the fine map on its held-out views (validation settings: no jitter, no noise)."""
    import torch
    from nerf_kinematics_amd.datasets import RaySampler
    from nerf_kinematics_amd.metrics import mse2psnr
    from nerf_kinematics_amd.trainer import Trainer, TrainOpts
    opts = TrainOpts(num_coarse=64, num_fine=128, num_random_rays=rays, perturb=True, radiance_field_noise_std=0.2,
                     lr=5e-4, seed=seed, act_format=act_format)
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


def comm_report(tr, dt_rank, steps, world, backend, dev):
    """Where a multi-rank step's time went (every rank calls it; rank 0 uses the result): the
    all-reduce time per step from events around the collective(s) of Trainer.step, every rank's own
    step time, and the collective library's version and settings."""
    import torch
    import torch.distributed as dist
    ar_ms = [sum(b.elapsed_time(e) for b, e in pairs) for pairs in tr.comm_events]
    mine = torch.tensor([dt_rank / steps * 1e3, sum(ar_ms) / max(len(ar_ms), 1)], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(mine) for _ in range(world)]
    if backend == "gloo":           # gloo gathers host tensors
        every = [t.cpu() for t in every]
        dist.all_gather(every, mine.cpu())
    else:
        dist.all_gather(every, mine)
    ms = [float(t[0]) for t in every]
    ar = [float(t[1]) for t in every]
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:          # noqa: BLE001 (a build without the binding still reports the rest)
        ver = f"unknown ({type(e).__name__})"
    return {"backend": backend, "collective_library": "RCCL " + ver if backend == "nccl" else backend,
            "allreduce_ms": sum(ar) / world, "allreduce_ms_per_rank": ar,
            "ms_per_step_per_rank": ms, "ms_per_step_min": min(ms), "ms_per_step_max": max(ms),
            "collectives_per_step": len(tr.comm_events[0]) if tr.comm_events else 0,
            "split_allreduce": bool(tr._split_allreduce(tr.R)),
            "gradient_bytes": int(tr.grad.numel()) * 4,
            "env": {k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS",
                                                   "RCCL_MSCCL_ENABLE", "HSA_ENABLE_IPC_MODE_LEGACY",
                                                   "NK_OVERLAP_ALLREDUCE")},
            "note": "allreduce_ms: events on the compute stream before the collective is queued and after the "
                    "stream has been made to wait for it (includes waiting for the slowest rank)"}


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
                     radiance_field_noise_std=0.2, lr=5e-4, seed=42, act_format=a.act_format)
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
        tr.comm_events = []             # (begin, end) event pairs around the step's collective(s)
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        nklib.check(L.nk_set_stage_events(handles[i], n_ev))
        loss = tr.step(*sampler.next(a.warmup + i))
    torch.cuda.synchronize()
    dt_rank = time.perf_counter() - t0      # this rank's own clock, before it waits for the others
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    nklib.check(L.nk_set_stage_events(None, 0))
    comm = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        comm = comm_report(tr, dt_rank, a.steps, world, backend, dev)

    if rank == 0:
        samples = rays * 192 * a.steps * world
        value = samples / dt
        span = lambda x, y: float(np.mean([row[EV[x]].elapsed_time(row[EV[y]]) for row in ev]))
        ab, db = ACT_FORMATS[a.act_format]
        shared = rays * 192 <= (1 << 20)                        # weight-gradient products of both networks in one shared launch
        pts_f, pts_c = rays * 192.0, rays * 64.0
        pts = 0.5 * (pts_f + pts_c)                              # points per launch, averaged like the durations

        def view(ms, points, flop_per_pt, bytes_per_pt):
            """One kernel's two rooflines: FLOP and HBM bytes per launch over its mean duration (ms)."""
            tf = points * flop_per_pt / (ms * 1e-3) / 1e12
            gb = points * bytes_per_pt / (ms * 1e-3) / 1e9
            return {"ms": ms, "tflops": tf, "frac_mfma": tf / PEAK_BF16_TFLOPS, "gbs": gb, "frac_hbm": gb / PEAK_HBM_GBS,
                    "bytes_per_point": bytes_per_pt, "flop_per_point": flop_per_pt}

        fwd_bytes = ACT_FEATS * ab + 320 + 16
        bwd_bytes = DZ_FEATS * db + 320 + 16
        dw_bytes = GEMM_DZ_FEATS * db + GEMM_ACT_FEATS * ab
        per_kernel = {
            "k_mlp_fwd<256,128,4,true>": dict(view(0.5 * (span("fwd_f0", "fwd_f1") + span("fwd_c0", "fwd_c1")), pts,
                                                   FLOP_PER_EVAL_8x256, fwd_bytes),
                                              ms_fine=span("fwd_f0", "fwd_f1"), ms_coarse=span("fwd_c0", "fwd_c1")),
            "k_mlp_bwd<256,128,8>": dict(view(0.5 * (span("bwd_fchain0", "bwd_fchain1") + span("bwd_cchain0", "bwd_cchain1")),
                                              pts, FLOP_BWD_CHAIN_8x256, bwd_bytes),
                                         ms_fine=span("bwd_fchain0", "bwd_fchain1"),
                                         ms_coarse=span("bwd_cchain0", "bwd_cchain1")),
        }
        if shared:      # small batch: the products of both networks ran in ONE launch behind the coarse chain
            dw = dict(view(span("bwd_cchain1", "bwd_end"), pts_f + pts_c, FLOP_PER_EVAL_8x256, dw_bytes), launches_per_step=1)
        else:
            dw = dict(view(0.5 * (span("bwd_fchain1", "bwd_fgemm1") + span("bwd_cchain1", "bwd_end")), pts,
                           FLOP_PER_EVAL_8x256, dw_bytes),
                      ms_fine=span("bwd_fchain1", "bwd_fgemm1"), ms_coarse=span("bwd_cchain1", "bwd_end"), launches_per_step=2)
        per_kernel["k_dw_gemm + k_dw_reduce"] = dw
        heavy_ms = (per_kernel["k_mlp_fwd<256,128,4,true>"]["ms"] + per_kernel["k_mlp_bwd<256,128,8>"]["ms"]) * 2 + \
            dw["ms"] * dw["launches_per_step"]
        step_tflops = value * 3 * FLOP_PER_EVAL_8x256 * 256 / 192 / 1e12 / world
        res = {
            "metric": "ray-samples/sec (train)", "value": value, "unit": "ray-samples/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": a.scaling, "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"lego-teacher {a.res}x{a.res}, 64 coarse + 128 fine, 8x256 coarse+fine NeRF, "
                                   f"train step (fwd+bwd+Adam), {rays} rays/step/GPU",
                       "rays_per_step_per_gpu": rays, "num_coarse": 64, "num_fine": 128,
                       "mlp_evals_per_s": value / 192 * 256, "final_loss": float(loss),
                       "saved_tensor_format": a.act_format,
                       "step_path": "Trainer.step -> nk_render_rays_fwd / nk_render_rays_bwd"},
            # The time-dominant kernel: the weight-gradient launch (k_dw_gemm, all 11 products
            # dW = dZ^T X of a network, and the k_dw_reduce behind it).  Per point it does the forward's
            # 1 186 816 FLOP and reads every dZ tensor and every layer input once.
            "roofline": {"bound": "mfma", "kernel": "k_dw_gemm + k_dw_reduce (weight-gradient products of one network)",
                         "achieved": dw["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": dw["frac_mfma"],
                         # HBM bytes per launch were NOT counted in this run: the figure below is
                         # (WRITE_SIZE + 2 x FETCH_SIZE) per point of the separate rocprofv3 --pmc passes of
                         # this command kept under profiles/, scaled to this launch; null for formats without a pass
                         "traffic": (pts if not shared else pts_f + pts_c) * PMC_DW_BYTES_PER_POINT[a.act_format]
                         if PMC_DW_BYTES_PER_POINT.get(a.act_format) else None,
                         "traffic_source": PMC_SOURCE if PMC_DW_BYTES_PER_POINT.get(a.act_format) else None,
                         "kernel_ms": dw["ms"], "flop_per_launch": (pts if not shared else pts_f + pts_c) * FLOP_PER_EVAL_8x256,
                         "hbm": {"achieved": dw["gbs"], "unit": "GB/s", "peak": PEAK_HBM_GBS, "frac": dw["frac_hbm"],
                                 "measured_read_ceiling": MEASURED_HBM_READ_GBS,
                                 "frac_of_measured": dw["gbs"] / MEASURED_HBM_READ_GBS,
                                 "bytes_per_launch": (pts if not shared else pts_f + pts_c) * dw_bytes},
                         "per_kernel": per_kernel,
                         "heavy_kernels_ms_per_step": heavy_ms,
                         "step_train_tflops": step_tflops,
                         "step_frac_of_mfma_peak": step_tflops / PEAK_BF16_TFLOPS,
                         "step_hbm_gbs": rays * 256 * (fwd_bytes + bwd_bytes + dw_bytes) / (dt / a.steps) / 1e9},
        }
        if comm is not None:
            res["comm"] = comm
    if world == 1 and not a.no_extras:
        # the same step over a longer timed region
        n_long = max(a.long_steps, 1)
        t_long = timed_steps(tr, sampler, n_long, 0, first=a.warmup + a.steps)
        res["value_long"] = {"value": rays * 192 / t_long, "unit": "ray-samples/s", "ms_per_step": t_long * 1e3,
                             "steps": n_long}
        # whole-image inference: one 400x400 view, nerf.validation settings (k_mlp_fwd<...,false>)
        rb = scene.rays(int(scene.i_val[0]), dev)
        tr.render(rb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            tr.render(rb)
        torch.cuda.synchronize()
        t_img = (time.perf_counter() - t0) / 5
        tr.render(rb, opts={"skip_empty_eps": 1e-3})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            tr.render(rb, opts={"skip_empty_eps": 1e-3})
        torch.cuda.synchronize()
        t_skip = (time.perf_counter() - t0) / 5
        res["render_400x400_ms"] = {"value": t_img * 1e3, "unit": "ms", "rays": int(rb.shape[0]), "repeats": 5,
                                    "with_skip_empty_eps_1e-3_ms": t_skip * 1e3,
                                    "rays_per_s": rb.shape[0] / t_img,
                                    "mfma_frac": rb.shape[0] * 256 * FLOP_PER_EVAL_8x256 / t_img / 1e12 / PEAK_BF16_TFLOPS,
                                    "note": "Trainer.render: 64+128 samples, no jitter, no noise, nothing saved"}
        tr = None
        torch.cuda.empty_cache()
        if a.act_format == "bf16":
            # the same step with the experimental 8-bit saved-tensor formats (off by default; never `value`)
            res["value_8bit_formats"] = {}
            for fmt in ("fp8", "fp8bf8"):
                o8 = TrainOpts(num_coarse=64, num_fine=128, num_random_rays=rays, perturb=True,
                               radiance_field_noise_std=0.2, lr=5e-4, seed=42, act_format=fmt)
                tr8 = Trainer(nk.ARCH_RUN_NERF, nk.ARCH_RUN_NERF, o8, dev, rays_per_step=rays)
                t8 = timed_steps(tr8, RaySampler(scene, rays, seed=1234), 40, 5)
                res["value_8bit_formats"][fmt] = {"value": rays * 192 / t8, "unit": "ray-samples/s", "ms_per_step": t8 * 1e3,
                                                  "steps": 40, "warmup": 5}
                tr8 = None
                torch.cuda.empty_cache()
            res["value_8bit_formats"]["note"] = ("nk_render_opts.act_format: fp8 = layer inputs saved as e4m3, fp8bf8 = and dZ as "
                                                 "e5m2; forward outputs unchanged, weight gradient coarser (tests/test_gpu_fp8.py)")
        opts1k = TrainOpts(num_coarse=64, num_fine=128, num_random_rays=1024, perturb=True,
                           radiance_field_noise_std=0.2, lr=5e-4, seed=42, act_format=a.act_format)
        tr1k = Trainer(nk.ARCH_RUN_NERF, nk.ARCH_RUN_NERF, opts1k, dev, rays_per_step=1024)
        t1k = timed_steps(tr1k, RaySampler(scene, 1024, seed=1234), 300, 30)
        res["value_r1024"] = {"value": 1024 * 192 / t1k, "unit": "ray-samples/s", "ms_per_step": t1k * 1e3,
                              "rays_per_step": 1024, "steps": 300, "warmup": 30}
        tr1k = None
        p, train_s = heldout_psnr(nk, scene, dev, a.psnr_iters, 4096, seed=42, act_format=a.act_format)
        res["heldout_psnr"] = {"value": p, "unit": "dB", "iters": a.psnr_iters, "rays_per_step": 4096, "seed": 42,
                               "train_views": a.views, "heldout_views": 2, "train_seconds": train_s,
                               "note": "fresh 8x256 pair on the lego-teacher views, fine map, validation settings"}
    if world == 1 and not a.no_cpu_baseline:
        # Threads: the CPUs this process may run on, capped by the box's own OMP_NUM_THREADS (the GPU
        # boxes of this pool show the whole machine in os.cpu_count() but grant a share of it).
        host_cores = os.cpu_count() or 1
        usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host_cores
        cap = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
        threads = min(usable, cap) if cap > 0 else usable
        torch.set_num_threads(threads)
        from oracle import nerf_oracle as O        # the CPU restatement is only ever the baseline leg
        v, t = cpu_baseline(O, a.cpu_rays)
        res["cpu_baseline"] = {"value": v, "unit": "ray-samples/s", "cores": torch.get_num_threads(),
                               "kind": "port", "host_cores": host_cores, "usable_cores": usable,
                               "threads": torch.get_num_threads(),
                               "sample": f"{a.cpu_rays} of the step's {rays} rays x (64+128) samples, one training step "
                                         f"(fwd+bwd+Adam) of the torch-CPU fp32 restatement, median of 3 "
                                         f"after 1 warm-up, {t:.2f} s/step; threads = min(CPUs this process may "
                                         f"use, OMP_NUM_THREADS)"}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
