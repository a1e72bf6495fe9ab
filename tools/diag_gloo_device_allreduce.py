"""Diagnostic (dev tool, one bounded run): two ranks of the real Trainer on ONE GPU exchanging the
gradient with gloo on the DEVICE tensor (what dist.allreduce_sum_ did before it switched to a host
copy for gloo).  Every rank arms faulthandler so that a stall dumps all of its Python stacks, and
logs the iteration it is in.  Usage:  python tools/diag_gloo_device_allreduce.py [iters]"""
import faulthandler, os, socket, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def control_main(iters):
    """The same exchange with NO kernel of this library: torch ops only."""
    import torch, torch.distributed as dist
    rank = int(os.environ["RANK"])
    log = open(os.path.join(ROOT, "gpurun_out", f"diag_gloo_control_rank{rank}.log"), "w", buffering=1)
    faulthandler.enable(log)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    g = torch.zeros(169096, device=dev)
    for i in range(iters):
        faulthandler.dump_traceback_later(40, exit=True, file=log)
        log.write(f"iter {i}\n")
        g.normal_()
        x = torch.randn(2048, 2048, device=dev); g[:2048] += (x @ x).sum(0)
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        g.mul_(0.5)
    faulthandler.cancel_dump_traceback_later()
    log.write("done\n")
    dist.barrier(); dist.destroy_process_group()


def rank_main(iters):
    import torch, torch.distributed as dist
    rank = int(os.environ["RANK"])
    log = open(os.path.join(ROOT, "gpurun_out", f"diag_gloo_rank{rank}.log"), "w", buffering=1)
    faulthandler.enable(log)
    import nerf_kinematics_amd as nk
    from nerf_kinematics_amd import dist as nd
    from nerf_kinematics_amd.datasets import RaySampler, teacher_scene
    from nerf_kinematics_amd.trainer import Trainer, TrainOpts
    nd.init_from_env("gloo")
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)

    def device_allreduce(flat, group=None):          # the pre-fix behaviour
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return flat
    nd.allreduce_sum_ = device_allreduce
    opts = TrainOpts(num_coarse=64, num_fine=64, num_random_rays=512, lr=5e-3, seed=42)
    tr = Trainer(nk.ARCH_LEGO_CKPT, nk.ARCH_LEGO_CKPT, opts, dev)
    scene = teacher_scene(None, 4, 1, 48, dev)
    smp = RaySampler(scene, 512, seed=1, rank=rank, world=2)
    for i in range(iters):
        faulthandler.dump_traceback_later(40, exit=True, file=log)
        log.write(f"iter {i} step\n")
        tr.step(*smp.next(i))
        if i % 10 == 9:
            log.write(f"iter {i} sharded render\n")
            tr.render(scene.rays(int(scene.i_val[0]), dev), shard=True)
            log.write(f"iter {i} loss {float(tr.loss):.5f} in_sync {nd.params_in_sync(tr.params)}\n")
    faulthandler.cancel_dump_traceback_later()
    log.write("done\n")
    dist.barrier(); nd.shutdown()


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    if "RANK" in os.environ:
        (control_main if os.environ.get("NK_DIAG_CONTROL") else rank_main)(iters)
    else:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        ps = [subprocess.Popen([sys.executable, __file__, str(iters)],
                               env=dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                                        MASTER_PORT=str(port))) for r in range(2)]
        rcs = [p.wait(timeout=170) for p in ps]
        print("exit codes", rcs)
        sys.exit(max(abs(r) for r in rcs))
