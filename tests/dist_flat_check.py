"""Helper of tests/test_gpu_dist.py: run under `python -m torch.distributed.run --nproc-per-node 2` with RD_DIST_BACKEND=gloo on a
one-GPU box (both ranks share cuda:0).  Checks the flat-buffer data parallelism of the fused optimizer:
  (a) ranks that start from DIFFERENT parameters are made equal by the initial broadcast and stay bit-identical over 2 steps;
  (b) for one more backward pass, the packed + all-reduced flat buffer equals, slice by slice and bit for bit, the per-tensor
      all-reduce of the same local gradients (what DistributedDataParallel computes), and 1 / world is the returned scale;
  (c) TWO backward passes before one step (gradient accumulation): the exchanged buffer holds the all-reduce of the ACCUMULATED
      gradients (buckets sent during the first pass are re-sent), never the first pass alone;
  (d) a parameter whose gradient is None on ONE rank only: every rank still applies the same update to it (group-wide presence mask),
      parameters and per-parameter Adam step counts stay identical across ranks.
(Two separate training runs cannot be compared tightly: atomics order -> ReLU sign flips make gradients differ ~1e-2 run to run.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
    model, cfg, *_ = _build_pillarnet(128)
    sd = model.state_dict(); seeded_fill_(sd, seed=90 + rank); model.load_state_dict(sd)          # ranks start DIFFERENT
    model = model.to(dev).train()
    opt = build_optimizer(model, cfg.OPTIMIZATION)
    sched, _ = build_scheduler(opt, 100, 1, -1, cfg.OPTIMIZATION)
    run = D.data_parallel(model, opt, 0, mode="flat")
    assert run is model and opt.flat_grad is not None
    fn = model_fn_decorator()

    def fwd_bwd(it):
        batch = make_batch(batch_size=2, n_lidar=300, n_radar=700, n_boxes=10, grid=128, seed=D.shard_seed(rank, it))
        sched.step(it); opt.zero_grad()
        loss, tb, _ = fn(run, dict(batch))
        loss.backward()

    for it in range(2):
        fwd_bwd(it)
        opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in opt.params])
    lo, hi = flat.clone(), flat.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert float((hi - lo).abs().max()) == 0.0, "parameters differ between ranks after 2 steps"
    # (b) same local gradients through both exchange schemes
    fwd_bwd(2)
    ref = []
    for p in opt.params:
        g = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().clone().reshape(-1)
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        ref.append(g)
    ref = torch.cat(ref)
    buf, scale = opt.allreduce_gradients()
    assert scale == 1.0 / world
    assert torch.equal(buf, ref), f"flat buffer differs from the per-tensor all-reduce: max {float((buf - ref).abs().max())}"
    dist.barrier()
    nb = len(opt.buckets.ranges) if getattr(opt, "buckets", None) is not None else 0
    if getattr(opt, "buckets", None) is not None:          # RD_DDP_OVERLAP=1: the bucketed, overlapped exchange
        assert nb >= 2 and all(v == hi - lo for v, (lo, hi) in zip(opt.buckets.left, opt.buckets.ranges)), "bucket bookkeeping not reset"
    # (c) gradient accumulation: two backward passes, one exchange
    opt.zero_grad()
    for k in (3, 4):
        batch = make_batch(batch_size=2, n_lidar=300, n_radar=700, n_boxes=10, grid=128, seed=D.shard_seed(rank, k))
        loss, tb, _ = fn(run, dict(batch))
        loss.backward()
    ref = []
    for p in opt.params:
        g = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().clone().reshape(-1)
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        ref.append(g)
    ref = torch.cat(ref)
    buf, _ = opt.allreduce_gradients()
    assert torch.equal(buf, ref), f"accumulated gradients: flat buffer differs from the per-tensor all-reduce by {float((buf - ref).abs().max())}"
    # a backward pass that is abandoned (no step): zero_grad() must leave clean bookkeeping behind
    fwd_bwd(5)
    opt.zero_grad()
    if getattr(opt, "buckets", None) is not None:
        assert not opt._works and not any(opt.buckets.seen) and not opt.buckets.dirty
    # (d) one parameter without a gradient on rank 1 only
    fwd_bwd(6)
    victim = max(range(len(opt.params)), key=lambda i: opt.params[i].numel() if opt.params[i].dim() == 1 else -1)     # a BatchNorm / bias vector
    before = opt.params[victim].detach().clone()
    if rank == 1:
        opt.params[victim].grad = None
    opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in opt.params])
    lo, hi = flat.clone(), flat.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert float((hi - lo).abs().max()) == 0.0, "a rank-local unused parameter made the ranks diverge"
    sk = opt.skipped_dev.float().clone()
    lo, hi = sk.clone(), sk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert float((hi - lo).abs().max()) == 0.0 and float(sk[victim]) == 0.0, "per-parameter Adam step counts differ between ranks"
    decay_only = before * (1.0 - opt.wd * opt.lr)
    assert float((opt.params[victim].detach() - decay_only).abs().max()) > 0.0, "the parameter was only decayed: the other rank's gradient was dropped"
    if rank == 0:
        print(f"DIST_FLAT_OK params {flat.numel()} buckets {nb} grad_norm {float((buf * scale).norm()):.3f}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
