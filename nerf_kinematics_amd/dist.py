"""Data-parallel helpers: rays shard across ranks, the only exchange is ONE all-reduce of the flat
gradient buffer that holds both networks (RCCL on GPUs via backend "nccl"; gloo in CPU tests).
The reference is single-GPU (no torch.distributed call site anywhere in /root/reference): data
parallelism over rays is a new capability."""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:     # NK_DIST_BACKEND=gloo: several ranks sharing one GPU (tests); RCCL wants one GPU per rank
            backend = os.environ.get("NK_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": torch.device(f"cuda:{local}")} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_rows(n: int, rank: int, world: int) -> slice:
    """Rows rank::world of a common batch (used by the 1-vs-N gradient equality test)."""
    return slice(rank, n, world)


def row_block(n: int, rank: int, world: int) -> slice:
    """The contiguous block of rows rank takes when n rows are rendered by `world` ranks (the last
    blocks may be shorter or empty)."""
    per = (n + world - 1) // world
    return slice(min(rank * per, n), min((rank + 1) * per, n))


def gather_rows(local: torch.Tensor, n: int, group=None) -> torch.Tensor:
    """Concatenate the row blocks of every rank (split as row_block does) on every rank: one
    all-gather of equally padded blocks, no other exchange.  Whole-image renders shard this way:
    rays are independent, so the hot loop needs no collective at all."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    per = (n + world - 1) // world
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    on_cpu = dist.get_backend(group) == "gloo" and pad.is_cuda       # gloo gathers host tensors
    src = pad.cpu() if on_cpu else pad
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src, group=group)
    out = torch.cat(parts)[:n]
    return out.to(local.device) if on_cpu else out


def allreduce_sum_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the flat gradient over ranks in place; the Adam kernel divides by world size."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if flat.is_cuda and dist.get_backend(group) == "gloo":
            # gloo is the CPU backend (tests with several ranks on one GPU): reduce a host copy.
            # Its own handling of device tensors has been seen to hang after a few iterations.
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)    # RCCL: in place, on the device
    return flat


def allreduce_sum_start(flat: torch.Tensor, group=None, force: bool = False):
    """Start summing `flat` over ranks in place and return a handle for allreduce_wait (None when
    the sum is already complete).  RCCL: the collective runs on its own stream behind everything
    queued on the current stream so far, so kernels launched afterwards overlap it.  ``force``
    issues the collective on a one-rank group too (tests of the stream ordering)."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_world_size(group) == 1 and not force:
        return None
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        allreduce_sum_(flat, group)         # host copy, synchronous (tests with several ranks on one GPU)
        return None
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)


def allreduce_wait(handle) -> None:
    """Make the current stream wait for a collective started by allreduce_sum_start."""
    if handle is not None:
        handle.wait()


def params_in_sync(flat: torch.Tensor, group=None) -> bool:
    """True when every rank holds bit-identical parameters (checked after optimiser steps)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return True
    host = flat.is_cuda and dist.get_backend(group) == "gloo"      # (see allreduce_sum_)
    lo, hi = (flat.cpu(), flat.cpu()) if host else (flat.clone(), flat.clone())
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool(torch.equal(lo, hi))


def shutdown() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
