"""Helper of tests/test_gpu_dist.py: run under `python -m torch.distributed.run --nproc-per-node 2` with RD_DIST_BACKEND=gloo on a
one-GPU box (both ranks share cuda:0).  --sync_bn of the reference (tools/train.py:144-145): every rank holds a DIFFERENT share of
the rows; with the BatchNorm layers synchronised, outputs, running statistics, input gradients and the SUM over ranks of the
parameter gradients must equal one process running the concatenated rows through the same (unsynchronised) kernels:
  (a) dense conv -> BN -> ReLU (+ residual) as one autograd node, (b) BatchNorm1d over sparse rows (different row counts per rank),
  (c) the dynamic pillar VFE (different point counts per rank), (d) one full distillation step of the converted model: finite loss,
      no deadlock, parameters stay identical across ranks."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn


def close(a, b, what, rtol=2e-4):
    a, b = a.detach().double(), b.detach().double()
    err = float((a - b).abs().max()); ref = float(b.abs().max())
    assert err <= rtol * max(ref, 1e-6), f"{what}: max err {err:.3e} vs max |ref| {ref:.3e}"


def summed(t):
    t = t.detach().clone()
    dist.all_reduce(t)
    return t


def main():
    from radardistill_amd import autograd as A
    from radardistill_amd import dense as D
    from radardistill_amd import dist as DD
    from radardistill_amd import kernels as K
    from radardistill_amd import sparse as SP
    from radardistill_amd.train import convert_sync_batchnorm
    from tests.seeded import seeded_fill_
    world, rank, _ = DD.env_world()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    DD.init_distributed(backend=os.environ.get("RD_DIST_BACKEND", "gloo"), device=dev)
    K.set_deterministic(True)
    g = np.random.default_rng(5)

    # ---- (a) dense conv + BN + residual + ReLU: rank r holds sample r of a batch of 2
    x_all = torch.from_numpy(g.normal(size=(2, 32, 16, 16)).astype(np.float32)).to(dev)
    res_all = torch.from_numpy(g.normal(size=(2 * 16 * 16, 64)).astype(np.float32)).to(dev)
    go_all = torch.from_numpy(g.normal(size=(2 * 16 * 16, 64)).astype(np.float32)).to(dev)

    def dense_case(sync, x, res, go):
        conv = nn.Conv2d(32, 64, 3, padding=1, bias=False); bn = nn.BatchNorm2d(64, eps=1e-3, momentum=0.01)
        m = nn.Sequential(conv, bn)
        sd = m.state_dict(); seeded_fill_(sd, seed=3); m.load_state_dict(sd)
        m = m.to(dev).train()
        if sync:
            m = convert_sync_batchnorm(m)
        else:
            A.SYNC_BN[0] = False
        x = x.clone().requires_grad_(True); res = res.clone().requires_grad_(True)
        A.begin_step(dev)
        out, *_ = D.conv_bn_act(x, m[0], m[1], residual_rows=res, act=1, return_rows=True)
        A.end_forward()
        (out * go).sum().backward()
        torch.cuda.synchronize()
        return out, x.grad, res.grad, m

    n = 16 * 16
    sl = slice(rank * n, (rank + 1) * n)
    out, gx, gres, m = dense_case(True, x_all[rank:rank + 1], res_all[sl], go_all[sl])
    assert isinstance(m[1], nn.SyncBatchNorm)
    grads = {k: summed(p.grad) for k, p in m.named_parameters()}
    r_out, r_gx, r_gres, r_m = dense_case(False, x_all, res_all, go_all)
    close(out, r_out[sl], "dense sync-BN output"); close(gx, r_gx[rank:rank + 1], "dense grad x"); close(gres, r_gres[sl], "dense grad residual")
    for k, p in r_m.named_parameters():
        close(grads[k], p.grad, f"dense grad {k}")
    close(m[1].running_mean, r_m[1].running_mean, "running_mean", 1e-5); close(m[1].running_var, r_m[1].running_var, "running_var", 1e-5)
    assert int(m[1].num_batches_tracked) == 1

    # ---- (b) BatchNorm1d over sparse rows: 300 rows on rank 0, 500 on rank 1
    f_all = torch.from_numpy(g.normal(1.0, 2.0, size=(800, 64)).astype(np.float32)).to(dev)
    g_all = torch.from_numpy(g.normal(size=(800, 64)).astype(np.float32)).to(dev)
    rs = slice(0, 300) if rank == 0 else slice(300, 800)

    def rows_case(sync, f, go):
        bn = nn.BatchNorm1d(64, eps=1e-3, momentum=0.01)
        sd = bn.state_dict(); seeded_fill_(sd, seed=4); bn.load_state_dict(sd)
        bn = bn.to(dev).train()
        if sync:
            bn = convert_sync_batchnorm(bn)
        else:
            A.SYNC_BN[0] = False
        f = f.clone().requires_grad_(True)
        A.begin_step(dev)
        y = SP.apply_rowwise(bn, f)
        A.end_forward()
        (y * go).sum().backward()
        return y, f.grad, bn

    y, gf, bn = rows_case(True, f_all[rs], g_all[rs])
    gw, gb = summed(bn.weight.grad), summed(bn.bias.grad)
    r_y, r_gf, r_bn = rows_case(False, f_all, g_all)
    close(y, r_y[rs], "rows sync-BN output"); close(gf, r_gf[rs], "rows grad x")
    close(gw, r_bn.weight.grad, "rows grad gamma"); close(gb, r_bn.bias.grad, "rows grad beta")
    close(bn.running_var, r_bn.running_var, "rows running_var", 1e-5)

    # ---- (c) dynamic pillar VFE: rank 0 holds sample 0 (700 points), rank 1 sample 1 (1100 points)
    from radardistill_amd.pcdet.config import AttrDict
    from radardistill_amd.pcdet.models.backbones_3d.vfe import __all__ as VFE
    from radardistill_amd.synthetic import bench_geometry
    pc_range, voxel, gs = bench_geometry(128)

    def pts(nn_, b, seed):
        r = np.random.default_rng(seed)
        p = np.zeros((nn_, 7), dtype=np.float32)
        p[:, 0] = b
        p[:, 1] = r.uniform(pc_range[0], pc_range[3], nn_); p[:, 2] = r.uniform(pc_range[1], pc_range[4], nn_)
        p[:, 3] = r.uniform(-1, 1, nn_); p[:, 4:] = r.normal(size=(nn_, 3))
        return torch.from_numpy(p).to(dev)

    def vfe_case(sync, points, B, go_fn):
        v = VFE["Radar_DynamicPillarVFESimple2D"](AttrDict(USE_NORM=True, WITH_DISTANCE=False, USE_ABSLOTE_XYZ=True, NUM_FILTERS=[32]),
                                                  num_point_features=6, voxel_size=voxel, grid_size=gs, point_cloud_range=pc_range)
        sd = v.state_dict(); seeded_fill_(sd, seed=6); v.load_state_dict(sd)
        v = v.to(dev).train()
        if sync:
            v = convert_sync_batchnorm(v)
        else:
            A.SYNC_BN[0] = False
        A.begin_step(dev)
        bd = v({"radar_points": points, "batch_size": B})
        A.end_forward()
        feats, coords = bd["radar_pillar_features"], bd["radar_pillar_coords"]
        (feats * go_fn(coords)).sum().backward()
        return feats, coords, v

    def go_fn(coords):          # a gradient that depends only on the pillar's cell, so both layouts see the same values
        c = coords.float()
        return torch.sin(c[:, 1:2] * 0.37 + c[:, 2:3] * 0.11 + torch.arange(32, device=dev).float() * 0.05)

    mine = pts(700, 0, 11) if rank == 0 else pts(1100, 0, 12)
    feats, coords, v = vfe_case(True, mine, 1, go_fn)
    vg = {k: summed(p.grad) for k, p in v.named_parameters()}
    both = torch.cat([pts(700, 0, 11), pts(1100, 1, 12)])
    r_feats, r_coords, r_v = vfe_case(False, both, 2, lambda c: go_fn(c))
    sel = r_coords[:, 0] == rank
    assert torch.equal(r_coords[sel][:, 1:], coords[:, 1:])
    close(feats, r_feats[sel], "VFE sync-BN features")
    for k, p in r_v.named_parameters():
        close(vg[k], p.grad, f"VFE grad {k}", 5e-4)
    close(v.pfn_layers[0].norm.running_var, r_v.pfn_layers[0].norm.running_var, "VFE running_var", 1e-5)
    K.set_deterministic(False)

    # ---- (d) the whole model, converted, one optimizer step on 2 ranks
    from radardistill_amd.pcdet.models import model_fn_decorator
    from radardistill_amd.synthetic import make_batch
    from radardistill_amd.train import build_optimizer, build_scheduler
    from tests.test_gpu_model import _build_pillarnet
    model, cfg, *_ = _build_pillarnet(128)
    sd = model.state_dict(); seeded_fill_(sd, seed=90); model.load_state_dict(sd)
    model = convert_sync_batchnorm(model.to(dev)).train()
    n_sync = sum(isinstance(mm, nn.SyncBatchNorm) for mm in model.modules())
    assert n_sync > 50, n_sync
    opt = build_optimizer(model, cfg.OPTIMIZATION)
    sched, _ = build_scheduler(opt, 100, 1, -1, cfg.OPTIMIZATION)
    run = DD.data_parallel(model, opt, 0, mode="flat")
    fn = model_fn_decorator()
    for it in range(2):
        batch = make_batch(batch_size=2, n_lidar=300, n_radar=700, n_boxes=10, grid=128, seed=DD.shard_seed(rank, it))
        sched.step(it); opt.zero_grad()
        loss, tb, _ = fn(run, dict(batch))
        loss.backward()
        opt.step()
        assert np.isfinite(float(loss.detach())), float(loss.detach())
    flat = torch.cat([p.detach().reshape(-1) for p in opt.params])
    lo, hi = flat.clone(), flat.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert float((hi - lo).abs().max()) == 0.0, "parameters differ between ranks after 2 synchronised-BN steps"
    # running statistics of a student BatchNorm are group-wide values, hence identical on both ranks
    rv = model.radar_backbone_2d.state_dict()
    k0 = [k for k in rv if k.endswith("running_var")][0]
    a, b = rv[k0].clone(), rv[k0].clone()
    dist.all_reduce(a, op=dist.ReduceOp.MIN); dist.all_reduce(b, op=dist.ReduceOp.MAX)
    assert float((b - a).abs().max()) == 0.0, "synchronised running_var differs between ranks"
    A.SYNC_BN[0] = False
    dist.barrier()
    if rank == 0:
        print("DIST_SYNCBN_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
