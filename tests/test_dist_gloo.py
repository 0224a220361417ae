"""World-size-2 gloo test (CPU) of the data-parallel glue used by bench.py / training for N > 1: process-group setup from
the torchrun environment, per-rank sample sharding, DDP gradient averaging with frozen parameters excluded, max-over-ranks
timing and the single-collective logging reduction."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from radardistill_amd import dist as D
    from radardistill_amd.synthetic import make_batch
    import torch.distributed as dist
    w, r, lr = D.init_distributed(backend="gloo")
    assert (w, r) == (world, rank)
    # sample sharding: distinct synthetic samples per rank
    b = make_batch(batch_size=1, n_lidar=50, n_radar=20, n_boxes=2, grid=128, seed=D.shard_seed(rank, 0))
    csum = float(b["radar_points"].sum())
    # DDP: frozen "teacher" + trainable "student"; gradients are averaged over ranks, frozen params untouched
    torch.manual_seed(0)
    net = torch.nn.ModuleDict({"teacher": torch.nn.Linear(4, 4), "student": torch.nn.Linear(4, 2)})
    for p in net["teacher"].parameters():
        p.requires_grad = False

    class M(torch.nn.Module):
        def __init__(self, net):
            super().__init__(); self.net = net

        def forward(self, x):
            with torch.no_grad():
                t = self.net["teacher"](x)
            return self.net["student"](t)

    m = D.wrap_ddp(M(net))
    x = torch.full((3, 4), float(rank + 1))
    m(x).sum().backward()
    g = net["student"].weight.grad.clone()
    # flat-all-reduce mode starts by making every rank equal to rank 0 (parameters AND buffers, mixed dtypes)
    bn = torch.nn.BatchNorm1d(3)
    with torch.no_grad():
        bn.weight.fill_(float(rank + 5)); bn.running_mean.fill_(float(rank + 7)); bn.num_batches_tracked.fill_(rank + 9)
    D.broadcast_parameters(bn, 0)
    assert float(bn.weight[0]) == 5.0 and float(bn.running_mean[2]) == 7.0 and int(bn.num_batches_tracked) == 9
    t_max = D.max_over_ranks(0.1 * (rank + 1))
    avg = D.average_scalars([float(rank), 10.0 * rank, 1.0])
    D.barrier()
    q.put((rank, csum, g.numpy(), t_max, avg, net["teacher"].weight.grad is None))
    dist.destroy_process_group()


def test_world_size_2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, c0, g0, t0, a0, f0), (r1, c1, g1, t1, a1, f1) = res
    assert c0 != c1                                         # ranks got different samples
    np.testing.assert_allclose(g0, g1)                      # DDP left identical (averaged) gradients on both ranks
    # expected: mean over ranks of d/dW sum(W t): rows of t summed -> 3 * teacher(x_rank)
    torch.manual_seed(0)
    teacher = torch.nn.Linear(4, 4)
    exp = sum(3 * teacher(torch.full((1, 4), float(r + 1))).detach()[0] for r in range(2)) / 2
    np.testing.assert_allclose(g0, np.tile(exp.numpy(), (2, 1)), rtol=1e-5)
    assert abs(t0 - 0.2) < 1e-9 and abs(t1 - 0.2) < 1e-9     # max over ranks
    assert a0 == a1 == [0.5, 5.0, 1.0]
    assert f0 and f1                                         # frozen parameters received no gradient


def _bucket_worker(rank, world, port, q):
    """GradBuckets on CPU tensors: hooks count buckets down during backward, each complete bucket is packed and all-reduced (async)
    while backward continues; the result must equal the per-tensor all-reduce bit for bit and the bookkeeping must reset."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from radardistill_amd import dist as D
    import torch.distributed as dist
    D.init_distributed(backend="gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.ReLU(), torch.nn.Linear(300, 300), torch.nn.ReLU(), torch.nn.Linear(300, 7))
    unused = torch.nn.Parameter(torch.zeros(11))                       # never receives a gradient: its bucket is closed by finish
    params = [unused] + list(net.parameters())                         # parameter order = forward order: the unused one sits in the LAST bucket
    numels = [p.numel() for p in params]
    offs = np.cumsum([0] + numels)
    flat = torch.zeros(int(offs[-1]))
    buckets = D.GradBuckets(numels, bucket_bytes=9_000)
    assert len(buckets.ranges) >= 3 and sorted(i for lo, hi in buckets.ranges for i in range(lo, hi)) == list(range(len(params)))
    works, order = [], []

    def launch(b):
        for w in works:                                                # (a re-launched dirty bucket re-packs a slice a collective may still own)
            w.wait()
        lo, hi = buckets.ranges[b]
        for i in range(lo, hi):
            g = params[i].grad
            flat[offs[i]:offs[i + 1]] = g.reshape(-1) if g is not None else 0.0
        works.append(dist.all_reduce(flat[offs[lo]:offs[hi]], async_op=True))
        order.append(b)

    for i, p in enumerate(params):
        p.register_post_accumulate_grad_hook(lambda _p, i=i: (lambda b: launch(b) if b is not None else None)(buckets.ready(i)))
    ok = True
    for it in range(2):
        for p in params:
            p.grad = None
        works.clear(); order.clear()
        x = torch.randn(5, 40, generator=torch.Generator().manual_seed(10 * rank + it))
        net(x).square().sum().backward()
        in_backward = list(order)                                      # buckets that started before backward returned
        for b in buckets.open_buckets():
            launch(b)
        for w in works:
            w.wait()
        buckets.reset()
        ref = []
        for p in params:
            g = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().clone().reshape(-1)
            dist.all_reduce(g)
            ref.append(g)
        ok = ok and torch.equal(flat, torch.cat(ref)) and len(in_backward) >= 2 and in_backward[0] == 0
    # gradient accumulation: TWO backward passes before the exchange is finished.  The second pass changes gradients whose buckets
    # were already sent: they must be marked dirty and sent again with the accumulated values (never silently dropped)
    for p in params:
        p.grad = None
    works.clear(); order.clear()
    for k in range(2):
        x = torch.randn(5, 40, generator=torch.Generator().manual_seed(100 + 10 * rank + k))
        net(x).square().sum().backward()
    first_pass = list(order)
    assert buckets.dirty == set(first_pass) and len(first_pass) >= 2
    for b in buckets.open_buckets():
        launch(b)
    for w in works:
        w.wait()
    assert sorted(order[len(first_pass):]) == list(range(len(buckets.ranges))) and not buckets.dirty
    buckets.reset()
    ref = []
    for p in params:
        g = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().clone().reshape(-1)
        dist.all_reduce(g)
        ref.append(g)
    ok = ok and torch.equal(flat, torch.cat(ref))
    q.put((rank, ok, float(flat.abs().sum())))
    dist.destroy_process_group()


def test_bucketed_overlapped_allreduce_equals_per_tensor_allreduce_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1]
    assert res[0][2] == res[1][2] and res[0][2] > 0
