"""Training step around the HIP render path: render_rays forward, img2mse, backward, gradient
all-reduce (RCCL when world_size > 1), Adam with the reference's exponential learning-rate decay,
re-pack of the bf16 weight image.

The loop structure and every constant follow the reference's experiment files
(/root/reference/trained_models/nerf-pytorch/lego-lowres3/config.yml): ``experiment`` block
(:10-17, train_iters / validate_every / save_every / randomseed), ``nerf.train`` block (:46-54),
``optimizer`` and ``scheduler`` (:64-69).  The learning rate is
``lr0 * lr_decay_factor ** (iter / (lr_decay * 1000))`` which reproduces the value stored in both
vendored checkpoints.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch

from . import lib as _lib
from .model import Arch, NeRFModel
from .render import make_opts, _ptr, _stream


@dataclass
class TrainOpts:
    """``nerf.train`` + ``optimizer`` + ``scheduler`` of a reference config.yml."""
    num_coarse: int = 64
    num_fine: int = 128
    num_random_rays: int = 1024
    perturb: bool = True
    lindisp: bool = False
    radiance_field_noise_std: float = 0.2
    white_background: bool = False
    lr: float = 5e-3
    lr_decay: int = 250
    lr_decay_factor: float = 0.1
    betas: tuple = (0.9, 0.999)
    eps: float = 1e-8
    seed: int = 42
    # format of the tensors saved for backward (experiment, no counterpart in the reference):
    # "bf16" (default), "fp8" = e4m3 activations, "fp8bf8" = e4m3 activations and e5m2 dZ
    act_format: str = "bf16"

    @staticmethod
    def from_config(cfg: dict) -> "TrainOpts":
        t = cfg.get("nerf", {}).get("train", {})
        o, s = cfg.get("optimizer", {}), cfg.get("scheduler", {})
        return TrainOpts(num_coarse=int(t.get("num_coarse", 64)), num_fine=int(t.get("num_fine", 128)),
                         num_random_rays=int(t.get("num_random_rays", 1024)), perturb=bool(t.get("perturb", True)),
                         lindisp=bool(t.get("lindisp", False)),
                         radiance_field_noise_std=float(t.get("radiance_field_noise_std", 0.0)),
                         white_background=bool(t.get("white_background", False)), lr=float(o.get("lr", 5e-4)),
                         lr_decay=int(s.get("lr_decay", 250)), lr_decay_factor=float(s.get("lr_decay_factor", 0.1)),
                         seed=int(cfg.get("experiment", {}).get("randomseed", 42)),
                         act_format=str(t.get("act_format", "bf16")))


def lr_at(iteration: int, o: TrainOpts) -> float:
    return o.lr * (o.lr_decay_factor ** (iteration / (o.lr_decay * 1000)))


class Trainer:
    """Owns both networks (one flat parameter / gradient / Adam-state buffer for the pair, so a
    data-parallel step is ONE all-reduce) and a workspace sized for ``rays_per_step`` rays."""

    def __init__(self, arch_coarse: Arch, arch_fine: Optional[Arch], opts: TrainOpts, device,
                 rays_per_step: Optional[int] = None, init_seed: Optional[int] = None,
                 process_group=None):
        self.L = _lib.load()
        self.opts = opts
        self.device = torch.device(device)
        self.R = int(rays_per_step or opts.num_random_rays)
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if (
            torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
        seed = opts.seed if init_seed is None else init_seed
        use_fine = opts.num_fine > 0 and arch_fine is not None
        nc = arch_coarse.num_params()
        nf = arch_fine.num_params() if use_fine else 0
        nc_pad = (nc + 63) // 64 * 64          # keep the fine block 256-byte aligned
        self.params = torch.zeros(nc_pad + nf, dtype=torch.float32, device=self.device)
        self.params[:nc] = _init_flat(arch_coarse, seed).to(self.device)
        if use_fine:
            self.params[nc_pad:] = _init_flat(arch_fine, seed + 1).to(self.device)
        # gradient and (behind it, 256-byte aligned) the loss scalar share one buffer: one fill per step
        n_pad = (nc_pad + nf + 63) // 64 * 64
        self._grad_loss = torch.zeros(n_pad + 64, dtype=torch.float32, device=self.device)
        self.grad = self._grad_loss[:nc_pad + nf]
        self.loss = self._grad_loss[n_pad:n_pad + 1]
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.coarse = NeRFModel(arch_coarse, self.params[:nc])
        self.fine = NeRFModel(arch_fine, self.params[nc_pad:]) if use_fine else None
        self._nc, self._nc_pad = nc, nc_pad
        # Two backward calls with an all-reduce per network, the fine one overlapped with the coarse
        # backward (_split_allreduce).  Off unless asked for (attribute, or NK_OVERLAP_ALLREDUCE=1): one
        # backward call and ONE all-reduce of the flat buffer is the default data-parallel step.
        self.overlap_allreduce: bool = os.environ.get("NK_OVERLAP_ALLREDUCE", "0") == "1"
        self.force_collective = False       # tests: issue the collectives on a one-rank group too
        # measurement hook: when a list, every step appends its [(begin, end), ...] event pairs recorded
        # on the compute stream around the gradient collective(s)
        self.comm_events: Optional[list] = None
        self.iteration = 0          # training iterations done (drives the learning-rate decay and the RNG)
        self.opt_steps = 0          # optimiser steps done (Adam bias correction; survives resume)
        self._alloc(self.R)

    # -- buffers ----------------------------------------------------------------------------
    def _alloc(self, R: int) -> None:
        o = self.opts
        self.ropts = make_opts(o.num_coarse, o.num_fine if self.fine is not None else 0, o.perturb, o.lindisp,
                               o.radiance_field_noise_std, o.white_background, o.seed, act_format=o.act_format)
        fc = self.fine._c if self.fine is not None else None
        self.ws_bytes = _lib.check(self.L.nk_render_workspace_bytes(self.coarse._c, fc, self.ropts, R, 1))
        self.ws = None              # allocated on the first non-staged step
        mk = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.device)
        self.rgb_c, self.rgb_f = mk(R, 3), (mk(R, 3) if self.fine is not None else None)
        self.d_rgb_c, self.d_rgb_f = mk(R, 3), (mk(R, 3) if self.fine is not None else None)
        self.out = _lib.NkRenderOut(_ptr(self.rgb_f), None, None, None, _ptr(self.rgb_c), None, None, None)

    def grad_views(self):
        gc = self.grad[:self._nc]
        gf = self.grad[self._nc_pad:] if self.fine is not None else None
        return gc, gf

    # -- one optimisation step -----------------------------------------------------------------
    def step(self, ray_batch: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """ray_batch [R,11], target [R,3] on this GPU; returns the loss tensor (device, no sync).

        The whole step is nk_render_rays_fwd, nk_mse_loss, nk_render_rays_bwd, (all-reduce),
        nk_adam_step and the re-pack of the bf16 weight images, all on the current stream.  To time
        its stages, hand HIP events to ``nk_set_stage_events`` before calling."""
        L, dev = self.L, self.device
        R = ray_batch.shape[0]
        if R != self.R:
            raise ValueError(f"Trainer was sized for {self.R} rays per step, got {R}")
        st = _stream(dev)
        self.ropts.seed = (self.opts.seed + 0x9E3779B1 * (self.iteration + 1) + 7919 * self._rank()) & 0xFFFFFFFF
        fc = self.fine._c if self.fine is not None else None
        fp = self.fine.packed.data_ptr() if self.fine is not None else None
        if self.ws is None:
            self.ws = torch.empty(self.ws_bytes + 256, dtype=torch.uint8, device=self.device)
            self.ws_ptr = (self.ws.data_ptr() + 255) & ~255
        with torch.cuda.device(dev):
            _lib.check(L.nk_render_rays_fwd(self.coarse._c, self.coarse.packed.data_ptr(), fc, fp,
                                            ray_batch.data_ptr(), R, self.ropts, 1, self.ws_ptr, self.ws_bytes,
                                            self.out, st))
            self._grad_loss.zero_()                             # gradient and loss in one fill
            _lib.check(L.nk_mse_loss(self.rgb_c.data_ptr(), _ptr(self.rgb_f), target.data_ptr(), R, 1.0,
                                     self.loss.data_ptr(), self.d_rgb_c.data_ptr(), _ptr(self.d_rgb_f), st))
            gc, gf = self.grad_views()
            bwd = lambda d_f, d_c: _lib.check(L.nk_render_rays_bwd(
                self.coarse._c, self.coarse.packed.data_ptr(), fc, fp, ray_batch.data_ptr(), R, self.ropts, self.ws_ptr,
                self.ws_bytes, d_f, d_c, gc.data_ptr(), _ptr(gf), st))
            timed = []

            def mark():
                if self.comm_events is None:
                    return None
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                return e
            if self._split_allreduce(R):
                # the fine network's gradient is complete before the coarse backward starts: its half
                # of the flat buffer is reduced while the coarse chain and products run
                from . import dist as nkdist
                bwd(_ptr(self.d_rgb_f), None)
                h_f = nkdist.allreduce_sum_start(self.grad[self._nc_pad:], self.pg, self.force_collective)
                bwd(None, self.d_rgb_c.data_ptr())
                b = mark()          # what the step still waits for once the coarse backward is queued
                h_c = nkdist.allreduce_sum_start(self.grad[:self._nc_pad], self.pg, self.force_collective)
                nkdist.allreduce_wait(h_f); nkdist.allreduce_wait(h_c)
                timed.append((b, mark()))
            else:
                bwd(_ptr(self.d_rgb_f), self.d_rgb_c.data_ptr())
                if self.world > 1:
                    from . import dist as nkdist
                    b = mark()
                    nkdist.allreduce_sum_(self.grad, self.pg)       # one flat buffer, both networks
                    timed.append((b, mark()))
            if self.comm_events is not None:
                self.comm_events.append(timed)
            self.iteration += 1
            self.opt_steps += 1
            _lib.check(L.nk_adam_step(self.params.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(),
                                      self.exp_avg_sq.data_ptr(), self.params.numel(), self.opt_steps,
                                      lr_at(self.iteration - 1, self.opts), self.opts.betas[0], self.opts.betas[1],
                                      self.opts.eps, 1.0 / self.world, st))
            self.coarse.pack()
            if self.fine is not None:
                self.fine.pack()
        return self.loss

    def _split_allreduce(self, R: int) -> bool:
        """Two backward calls with an all-reduce per network (the fine network's overlapped with the
        coarse backward) instead of one call and one all-reduce: only when ``overlap_allreduce`` asks
        for it.  The flat gradient of the 8x256 pair is 4.77 MB, tens of microseconds on xGMI against a
        step of milliseconds, so the overlap buys well under 1 % and the simpler step is the default."""
        return self.fine is not None and bool(self.overlap_allreduce)

    def _rank(self) -> int:
        return torch.distributed.get_rank(self.pg) if self.world > 1 else 0

    # -- checkpoints ----------------------------------------------------------------------------
    def export_state(self, loss: float = float("nan"), psnr: float = float("nan")):
        """Everything a reference-layout checkpoint carries, copied to the CPU.  ``iter`` is the index
        of the last completed iteration, as in the reference's files (its checkpoint110000 holds
        Adam step counts of 110001 and more)."""
        from .checkpoint import TrainState
        nc, ncp = self._nc, self._nc_pad
        cpu = lambda t: t.detach().to("cpu", copy=True)
        fine = self.fine is not None
        cut = lambda t: [cpu(t[:nc])] + ([cpu(t[ncp:])] if fine else [])
        return TrainState(iteration=self.iteration - 1, coarse=cpu(self.params[:nc]),
                          fine=cpu(self.params[ncp:]) if fine else None, exp_avg=cut(self.exp_avg),
                          exp_avg_sq=cut(self.exp_avg_sq), opt_step=self.opt_steps,
                          lr=lr_at(max(self.iteration - 1, 0), self.opts), loss=loss, psnr=psnr)

    def import_state(self, s) -> None:
        """Resume from a TrainState (parameters, Adam moments); training continues with the iteration
        after the stored one."""
        nc, ncp = self._nc, self._nc_pad
        if s.coarse.numel() != nc or ((s.fine is not None) != (self.fine is not None)):
            raise ValueError("checkpoint networks do not match the trainer's")
        self.params[:nc] = s.coarse.to(self.device)
        if self.fine is not None:
            self.params[ncp:] = s.fine.to(self.device)
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        if s.exp_avg:
            self.exp_avg[:nc] = s.exp_avg[0].to(self.device); self.exp_avg_sq[:nc] = s.exp_avg_sq[0].to(self.device)
            if self.fine is not None:
                self.exp_avg[ncp:] = s.exp_avg[1].to(self.device); self.exp_avg_sq[ncp:] = s.exp_avg_sq[1].to(self.device)
        self.iteration = int(s.iteration) + 1
        self.opt_steps = int(s.opt_step) if s.exp_avg else 0
        self.coarse.pack()
        if self.fine is not None:
            self.fine.pack()

    # -- inference --------------------------------------------------------------------------
    @torch.no_grad()
    def render(self, ray_batch: torch.Tensor, chunk: int = 65536, opts: Optional[dict] = None,
               shard: bool = False) -> Dict[str, torch.Tensor]:
        """Whole-image render with the ``nerf.validation`` settings (no jitter, no noise).

        ``shard=True`` (every rank of the process group must call it, with the same rays): each
        rank renders one contiguous block of the rays and the blocks are all-gathered, so every
        rank returns the whole image."""
        from .render import render_rays
        o = self.opts
        n_all = ray_batch.shape[0]
        if shard and self.world > 1:
            from . import dist as nkdist
            ray_batch = ray_batch[nkdist.row_block(n_all, self._rank(), self.world)]
        kw = dict(num_coarse=o.num_coarse, num_fine=o.num_fine if self.fine is not None else 0, perturb=False,
                  lindisp=o.lindisp, radiance_field_noise_std=0.0, white_background=o.white_background)
        if opts:
            kw.update(opts)
        parts = []
        for s in range(0, ray_batch.shape[0], chunk):
            parts.append(render_rays(ray_batch[s:s + chunk], self.coarse, self.fine, training=False, **kw))
        names = ("rgb_fine", "disp_fine", "acc_fine", "rgb_coarse", "disp_coarse", "acc_coarse")
        if not parts:                   # an empty block of a sharded render
            z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)
            out = {n: (z(0, 3) if n.startswith("rgb") else z(0)) for n in names}
        else:
            out = {n: torch.cat([p[i] for p in parts]) for i, n in enumerate(names)}
        if shard and self.world > 1:
            out = {n: nkdist.gather_rows(v, n_all, self.pg) for n, v in out.items()}
        return out


class GraphedStep:
    """The whole single-GPU training iteration -- pixel draw + get_rays + colours, render_rays
    forward, img2mse, backward, Adam, re-pack of the weight images -- captured ONCE into a HIP graph
    and replayed per iteration.  Everything that changes between iterations (seeds, pixel counter,
    view index, Adam's two step-dependent scalars) lives in a 32-byte ``nk_step_state`` in device
    memory that the kernels read when they run; per iteration the host queues one one-thread kernel
    that carries the struct as its argument and one graph launch, instead of ~40 kernel launches.  At the
    reference's own batch size (1 024 rays, 4x128 networks) the step is launch-bound, which is what
    this removes.  Same kernels, same arithmetic as ``Trainer.step``: the two give the same bits.

    ``capture=False`` issues the same launch sequence eagerly (what the graph is checked against)."""

    def __init__(self, trainer: "Trainer", sampler, capture: bool = True):
        import ctypes as C
        if trainer.world != 1:
            raise ValueError("GraphedStep: single-process training only (the all-reduce is not captured)")
        sc = sampler.scene
        if sampler.R != trainer.R:
            raise ValueError(f"sampler draws {sampler.R} rays, trainer was sized for {trainer.R}")
        self.tr, self.smp, self.L = trainer, sampler, trainer.L
        dev = trainer.device
        # camera table of every view and the image stack, on the device
        cams = (_lib.NkCamera * sc.images.shape[0])(*[sc.camera(k) for k in range(sc.images.shape[0])])
        self.cams = torch.frombuffer(bytearray(bytes(cams)), dtype=torch.uint8).to(dev)
        self.n_views = int(sc.images.shape[0])
        self.image_stride = int(sc.images.stride(0)) * sc.images.element_size()
        self.channels = int(sc.images.shape[-1])
        self.white = int(sc.white_background)
        self._state = _lib.NkStepState()
        self.state_dev = torch.zeros(C.sizeof(_lib.NkStepState), dtype=torch.uint8, device=dev)
        o = trainer.ropts
        self.ropts = _lib.NkRenderOpts(o.num_coarse, o.num_fine, o.perturb, o.lindisp, o.white_background,
                                       o.radiance_field_noise_std, 0, o.act_format, self.state_dev.data_ptr(), 0.0, 0)
        if trainer.ws is None:
            trainer.ws = torch.empty(trainer.ws_bytes + 256, dtype=torch.uint8, device=dev)
            trainer.ws_ptr = (trainer.ws.data_ptr() + 255) & ~255
        self.graph = None
        if capture:
            # capture on a side stream; the launches only record (nothing runs, nothing is advanced)
            self._fill_state()
            _lib.check(self.L.nk_set_step_state(self._state, self.state_dev.data_ptr(), _stream(dev)))
            torch.cuda.synchronize(dev)
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                # one eager pass of the kernels first would advance the parameters, so instead the
                # launch-time set-up that capture forbids (function attributes) is done by a throw-away
                # pass on scratch copies of everything the step writes
                keep = [t.clone() for t in (trainer.params, trainer.exp_avg, trainer.exp_avg_sq)]
                self._launch(side.cuda_stream)
                side.synchronize()
                for t, k in zip((trainer.params, trainer.exp_avg, trainer.exp_avg_sq), keep):
                    t.copy_(k)
                trainer.coarse.pack()
                if trainer.fine is not None:
                    trainer.fine.pack()
                side.synchronize()
                with torch.cuda.graph(g, stream=side):
                    self._launch(side.cuda_stream)
            torch.cuda.current_stream(dev).wait_stream(side)
            self.graph = g

    def _fill_state(self) -> None:
        tr, smp, st = self.tr, self.smp, self._state
        it = tr.iteration
        st.render_seed = (tr.opts.seed + 0x9E3779B1 * (it + 1) + 7919 * tr._rank()) & 0xFFFFFFFF
        st.pixel_seed = smp.seed
        st.pixel_counter = smp.counter_of(it)
        st.view = smp.view_of(it)
        step = tr.opt_steps + 1
        b1, b2 = tr.opts.betas
        st.adam_step_size = lr_at(it, tr.opts) / (1.0 - b1 ** step)
        st.adam_bc2_sqrt = math.sqrt(1.0 - b2 ** step)

    def _launch(self, st: int) -> None:
        """The launch sequence of one iteration on stream `st` (captured once, or issued eagerly)."""
        tr, L, smp = self.tr, self.L, self.smp
        R = tr.R
        fc = tr.fine._c if tr.fine is not None else None
        fp = tr.fine.packed.data_ptr() if tr.fine is not None else None
        sp = self.state_dev.data_ptr()
        _lib.check(L.nk_sample_training_rays_dev(self.cams.data_ptr(), self.n_views, smp.scene.images.data_ptr(),
                                                 self.image_stride, self.channels, self.white, sp, R,
                                                 smp.ray_batch.data_ptr(), smp.target.data_ptr(), smp.pixels.data_ptr(), st))
        rb = smp.ray_batch.data_ptr()
        _lib.check(L.nk_render_rays_fwd(tr.coarse._c, tr.coarse.packed.data_ptr(), fc, fp, rb, R, self.ropts, 1,
                                        tr.ws_ptr, tr.ws_bytes, tr.out, st))
        tr._grad_loss.zero_()
        _lib.check(L.nk_mse_loss(tr.rgb_c.data_ptr(), _ptr(tr.rgb_f), smp.target.data_ptr(), R, 1.0, tr.loss.data_ptr(),
                                 tr.d_rgb_c.data_ptr(), _ptr(tr.d_rgb_f), st))
        gc, gf = tr.grad_views()
        _lib.check(L.nk_render_rays_bwd(tr.coarse._c, tr.coarse.packed.data_ptr(), fc, fp, rb, R, self.ropts, tr.ws_ptr,
                                        tr.ws_bytes, _ptr(tr.d_rgb_f), tr.d_rgb_c.data_ptr(), gc.data_ptr(), _ptr(gf), st))
        _lib.check(L.nk_adam_step_dev(tr.params.data_ptr(), tr.grad.data_ptr(), tr.exp_avg.data_ptr(),
                                      tr.exp_avg_sq.data_ptr(), tr.params.numel(), sp, tr.opts.betas[0], tr.opts.betas[1],
                                      tr.opts.eps, 1.0, st))
        for m in (tr.coarse, tr.fine):
            if m is not None:
                _lib.check(L.nk_pack_weights(m._c, m.flat.data_ptr(), m.packed.data_ptr(), st))

    def step(self) -> torch.Tensor:
        """One iteration (the trainer's next); returns the loss tensor (device, no sync)."""
        tr = self.tr
        dev = tr.device
        with torch.cuda.device(dev):
            self._fill_state()       # the struct goes to the device as a kernel argument: no host buffer to race with
            _lib.check(self.L.nk_set_step_state(self._state, self.state_dev.data_ptr(), _stream(dev)))
            if self.graph is not None:
                self.graph.replay()
            else:
                self._launch(_stream(dev))
        tr.iteration += 1
        tr.opt_steps += 1
        return tr.loss


def _init_flat(arch: Arch, seed: int) -> torch.Tensor:
    """torch.nn.Linear default init, U(-1/sqrt(in), 1/sqrt(in)), in state-dict order."""
    g = torch.Generator().manual_seed(seed)
    parts = []
    for _, (o, i) in arch.param_shapes():
        b = 1.0 / math.sqrt(i)
        parts.append((torch.rand(o * i, generator=g) * 2 - 1) * b)
        parts.append((torch.rand(o, generator=g) * 2 - 1) * b)
    return torch.cat(parts)


def psnr_from_mse(mse: float) -> float:
    return -10.0 * math.log10(max(mse, 1e-12))
