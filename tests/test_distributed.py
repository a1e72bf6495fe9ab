"""World-size-2 gloo test of the data-parallel contract (CPU; the oracle stands in for the GPU
compute): summing per-rank gradients of the per-rank mean loss and dividing by the world size
equals the single-process gradient on the concatenated batch, and identical updates keep the
parameters bit-identical on every rank.  Also the row split / all-gather a sharded whole-image
render uses."""
import math
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nerf_oracle as O


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _batch(R=16):
    focal = 0.5 * 16 / math.tan(0.5 * 0.6911112070083618)
    ro, rd = O.get_rays(16, 16, focal, O.spherical_pose(45.0, -30.0, 4.0))
    rb = O.make_ray_batch(ro, rd, 2.0, 6.0).view(16, 16, 11)[6:10, 6:10].reshape(-1, 11)[:R].contiguous()   # rays through the object
    torch.manual_seed(1)
    return rb, torch.rand(R, 3)


def _weights():
    import numpy as np
    w = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lego_lowres3_weights.npz"))
    return torch.from_numpy(w["coarse"]), torch.from_numpy(w["fine"])     # trained: the rays hit density


def _grad(rb, target, flat_c, flat_f):
    fc, ff = flat_c.clone().requires_grad_(True), flat_f.clone().requires_grad_(True)
    a = O.ARCH_LEGO_CKPT
    out = O.render_rays(rb, O.NeRFNet.from_flat(a, fc), O.NeRFNet.from_flat(a, ff), O.RenderOpts(16, 16))
    O.nerf_loss(out, target).backward()
    return torch.cat([fc.grad, ff.grad])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from nerf_kinematics_amd import dist as nd
    r, w, _ = nd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    rb, target = _batch()
    fc, ff = _weights()
    sl = nd.shard_rows(rb.shape[0], rank, world)
    g = _grad(rb[sl], target[sl], fc, ff)
    nd.allreduce_sum_(g)
    g /= world
    params = torch.cat([fc, ff])
    m, v = torch.zeros_like(params), torch.zeros_like(params)
    O.adam_step(params, g, m, v, 1, 5e-3)
    ok = nd.params_in_sync(params)
    # row sharding of a whole-image render: blocks of 7 rows over 2 ranks (4 + 3), and an empty block
    full = torch.arange(21, dtype=torch.float32).view(7, 3)
    for n in (7, 1):
        blk = nd.row_block(n, rank, world)
        assert torch.equal(nd.gather_rows(full[:n][blk].contiguous(), n), full[:n])
    assert nd.row_block(7, 0, 2) == slice(0, 4) and nd.row_block(7, 1, 2) == slice(4, 7) and nd.row_block(1, 1, 2) == slice(1, 1)
    if rank == 0:
        q.put((g.numpy().copy(), ok))     # by value: the worker's shared memory dies with it
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_equals_single_process():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    g2, ok = q.get(timeout=300)
    g2 = torch.from_numpy(g2)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    rb, target = _batch()
    fc, ff = _weights()
    g1 = _grad(rb, target, fc, ff)
    assert ok
    assert float(g1.norm()) > 0
    assert float((g1 - g2).norm() / g1.norm()) < 1e-5


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: the script starts two ranks itself (gloo here,
    no GPU work with --launch-check) and rank 0 prints ONE JSON line that says n_gpus == 2."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["NK_DIST_BACKEND"] = "gloo"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["launch_check"] is True
    # under a launcher (WORLD_SIZE set) the script is one rank and starts nothing
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env1,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["n_gpus"] == 1
