"""RCCL smoke on however many GPUs are visible (dev tool): one process per GPU, backend nccl, the
collectives the trainer and bench.py use (async all-reduce in place on slices of one flat buffer,
MAX all-reduce, barrier, all-gather).  With one GPU this still loads RCCL and runs its kernels.
    python tools/rccl_smoke.py"""
import os, socket, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_main():
    import torch, torch.distributed as dist
    from nerf_kinematics_amd import dist as nd
    rank, world, local = nd.init_from_env("nccl")
    dev = torch.device(f"cuda:{local}"); torch.cuda.set_device(dev)
    if world == 1 and not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    flat = torch.arange(1000, dtype=torch.float32, device=dev) * (rank + 1)
    h1 = dist.all_reduce(flat[512:], op=dist.ReduceOp.SUM, async_op=True)
    flat[:512].mul_(2.0)                                  # compute on the current stream meanwhile
    h2 = dist.all_reduce(flat[:512], op=dist.ReduceOp.SUM, async_op=True)
    h1.wait(); h2.wait()
    tot = sum(range(1, world + 1))
    ref = torch.arange(1000, dtype=torch.float32) * tot
    ref[:512] *= 2.0
    assert torch.equal(flat.cpu(), ref), "all-reduce on slices of a flat buffer"
    t = torch.tensor([float(rank)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX); assert int(t.item()) == world - 1
    dist.barrier()
    out = nd.gather_rows(torch.full((3, 2), float(rank), device=dev), 3 * world)
    assert out.shape == (3 * world, 2)
    assert nd.params_in_sync(torch.ones(10, device=dev))
    if rank == 0:
        print(f"rccl smoke ok: world {world}, backend {dist.get_backend()}")
    dist.destroy_process_group()


if __name__ == "__main__":
    if "RANK" in os.environ:
        rank_main()
    else:
        import torch
        n = torch.cuda.device_count()
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        ps = [subprocess.Popen([sys.executable, __file__], env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                                                                  MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))) for r in range(n)]
        sys.exit(max(p.wait(timeout=170) for p in ps))
