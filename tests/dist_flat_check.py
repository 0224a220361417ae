"""Helper of tests/test_gpu_dist.py: run under `python -m torch.distributed.run --nproc-per-node 2` with RD_DIST_BACKEND=gloo on a
one-GPU box (both ranks share cuda:0).  Trains 2 steps with the flat-buffer all-reduce and with DistributedDataParallel from the
same initial state and rank-dependent batches, and checks (a) every rank ends with identical parameters, (b) the two exchange
schemes agree."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist


def main():
    from radardistill_amd import dist as D
    from radardistill_amd.pcdet.models import model_fn_decorator
    from radardistill_amd.synthetic import make_batch
    from radardistill_amd.train import build_optimizer, build_scheduler
    from tests.seeded import seeded_fill_
    from tests.test_gpu_model import _build_pillarnet
    world, rank, _ = D.env_world()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    D.init_distributed(backend=os.environ.get("RD_DIST_BACKEND", "gloo"), device=dev)
    finals = {}
    for mode in ("flat", "torch"):
        model, cfg, *_ = _build_pillarnet(128)
        sd = model.state_dict(); seeded_fill_(sd, seed=90 + rank if mode == "flat" else 90); model.load_state_dict(sd)   # flat: ranks start DIFFERENT, the broadcast must fix it
        if mode == "torch":
            sd = model.state_dict(); seeded_fill_(sd, seed=90); model.load_state_dict(sd)
        model = model.to(dev).train()
        opt = build_optimizer(model, cfg.OPTIMIZATION)
        sched, _ = build_scheduler(opt, 100, 1, -1, cfg.OPTIMIZATION)
        run = D.data_parallel(model, opt, 0, mode=mode)
        assert (opt.flat_grad is not None) == (mode == "flat")
        fn = model_fn_decorator()
        for it in range(2):
            batch = make_batch(batch_size=2, n_lidar=300, n_radar=700, n_boxes=10, grid=128, seed=D.shard_seed(rank, it))
            sched.step(it); opt.zero_grad()
            loss, tb, _ = fn(run, dict(batch))
            loss.backward()
            opt.step()
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters() if p.requires_grad])
        lo, hi = flat.clone(), flat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert float((hi - lo).abs().max()) == 0.0, f"{mode}: parameters differ between ranks"
        finals[mode] = flat
        del run, model, opt
    a, b = finals["flat"], finals["torch"]
    # rank 0 of the flat run started from seed 90 as well: identical data, identical math up to the summation order of the exchange
    rel = float((a - b).norm() / b.norm())
    assert rel < 1e-3, f"flat vs DDP parameters differ: rel L2 {rel}"
    dist.barrier()
    if rank == 0:
        print(f"DIST_FLAT_OK rel_l2={rel:.2e}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
