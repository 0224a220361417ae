"""GPU parity tests, part 2: DCNv2, AFD/PFD, rotated overlap, optimizer, the 2-D modules against the golden fixtures
generated from the reference's own modules, and the whole PillarNet distillation step against the CPU oracle."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import bev as obev, head as ohead, optim as ooptim, pillarnet as opn      # noqa: E402  (checker only)
from radardistill_amd.synthetic import bench_geometry, make_batch                       # noqa: E402
from tests.seeded import seeded_fill_                                                   # noqa: E402
from tests.test_gpu_kernels import close, DEV                                           # noqa: E402

BEV_CFG = dict(LAYER_NUMS=[5, 5], LAYER_STRIDES=[1, 2], NUM_FILTERS=[256, 256], UPSAMPLE_STRIDES=[1, 2], NUM_UPSAMPLE_FILTERS=[128, 128])


def _cl(x):
    return x.to(DEV).contiguous(memory_format=torch.channels_last)


def _bev_inputs(seed, B=1, S=16):
    g = np.random.default_rng(seed)
    x4 = g.normal(0, 1, size=(B, 256, S, S)).astype(np.float32)
    x4 *= (g.uniform(size=(B, 1, S, S)) < 0.4)
    x5 = g.normal(0, 1, size=(B, 256, S // 2, S // 2)).astype(np.float32)
    return torch.from_numpy(x4), torch.from_numpy(x5)


# ------------------------------------------------------------------------------------------ DCNv2
@pytest.mark.parametrize("columns", [True, False])
@pytest.mark.parametrize("C,Cout,H,W,stride", [(64, 64, 9, 8, 2), (32, 96, 7, 7, 1), (256, 256, 16, 16, 2)])
def test_dcn_forward_backward_vs_oracle(C, Cout, H, W, stride, columns, monkeypatch):
    """columns: the training path's form (rd_dcn_columns + plain GEMMs) / the sampling fused into the GEMM's operand staging (index mode 3)."""
    from radardistill_amd.pcdet.ops.basicblock import modulated_deform_conv as MDC
    from radardistill_amd.pcdet.ops.basicblock.modulated_deform_conv import ModulatedDeformConv
    monkeypatch.setattr(MDC, "DCN_COLUMNS", columns)
    rng = np.random.default_rng(C + H)
    B = 2
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    x = torch.from_numpy(rng.normal(size=(B, C, H, W)).astype(np.float32))
    off = torch.from_numpy((rng.normal(size=(B, 18, Ho, Wo)) * 1.5).astype(np.float32))
    off[0, :, 0, 0] = -3.0          # samples far outside the map
    off[1, 0::2, -1, -1] = 0.0      # exactly integer positions
    mask = torch.from_numpy(rng.uniform(0, 1, size=(B, 9, Ho, Wo)).astype(np.float32))
    m = ModulatedDeformConv(C, Cout, 3, stride, 1, bias=False)
    w, b = m.weight.detach().clone(), m.bias.detach().clone()
    xr, offr, mr, wr = [t.clone().requires_grad_(True) for t in (x, off, mask, w)]
    ref = obev.modulated_deform_conv(xr, offr, mr, wr, b, stride=stride, pad=1)
    go = torch.from_numpy(rng.normal(size=tuple(ref.shape)).astype(np.float32))
    (ref * go).sum().backward()
    md = m.to(DEV)
    xd, offd, maskd = [t.to(DEV).requires_grad_(True) for t in (x, off, mask)]
    out = md(xd, offd, maskd)
    close(out, ref, what="dcn fwd")
    (out * go.to(DEV)).sum().backward()
    close(xd.grad, xr.grad, atol=2e-4, what="dcn grad input")
    close(offd.grad, offr.grad, atol=3e-4, what="dcn grad offset")
    close(maskd.grad, mr.grad, atol=2e-4, what="dcn grad mask")
    close(md.weight.grad, wr.grad, atol=2e-4, what="dcn grad weight")
    assert md.bias.grad is None                       # bias=False: parameter exists, is added, but stays frozen


def test_dcn_zero_offset_unit_mask_is_conv2d():
    """Known answer restated from the reference's pcdet/ops/basicblock/test.py:69-110."""
    from radardistill_amd.pcdet.ops.basicblock.modulated_deform_conv import ModulatedDeformConv
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 64, 10, 12, generator=g)
    m = ModulatedDeformConv(64, 32, 3, 2, 1, bias=True)
    ref = F.conv2d(x, m.weight, m.bias, stride=2, padding=1)
    md = copy.deepcopy(m).to(DEV)
    out = md(x.to(DEV), torch.zeros(2, 18, 5, 6, device=DEV), torch.ones(2, 9, 5, 6, device=DEV))
    close(out, ref, rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------------------------ AFD / PFD
def _radar_distill(seed=13):
    from radardistill_amd.pcdet.config import AttrDict
    from radardistill_amd.pcdet.models.backbones_2d import __all__ as REG
    cfg = AttrDict(dict(BEV_CFG, VOXEL_SIZE=[0.2, 0.2, 8.0], POINT_CLOUD_RANGE=[-12.8, -12.8, -5, 12.8, 12.8, 3]))
    m = REG["Radar_Distill"](cfg, input_channels=256)
    sd = m.state_dict(); seeded_fill_(sd, seed=seed); m.load_state_dict(sd)
    return m.to(DEV)


def test_afd_pfd_golden_and_gradients(golden_dir):
    g = np.load(f"{golden_dir}/g3_radar_distill.npz")
    m = _radar_distill()
    r = np.random.default_rng(23)
    lid = torch.from_numpy(r.normal(0.2, 1, size=(2, 256, 16, 16)).astype(np.float32)) * \
        torch.from_numpy((r.uniform(size=(2, 1, 16, 16)) < 0.5).astype(np.float32))
    rad = torch.from_numpy(r.normal(0.0, 1, size=(2, 256, 16, 16)).astype(np.float32))
    f, ml = m.low_loss(_cl(lid), _cl(rad))
    close(f, g["afd_feature"], rtol=1e-3, atol=1e-6); close(ml, g["afd_mask"], rtol=1e-3, atol=1e-6)
    f2, _ = m.low_loss(_cl(lid.abs() + 1.0), _cl(rad))
    assert torch.isnan(f2).item() and np.isnan(g["afd_feature_nan"])                   # the reference's 0*inf edge case
    hms = [torch.from_numpy(r.uniform(0, 1, size=(2, c, 16, 16)).astype(np.float32) ** 6) for c in (1, 2, 2, 1, 2, 2)]
    logits = [torch.from_numpy(r.normal(-2.0, 1.5, size=(2, c, 16, 16)).astype(np.float32)) for c in (1, 2, 2, 1, 2, 2)]
    r1, r2, l1, l2 = [torch.from_numpy(r.normal(0, 1, size=(2, 256, 16, 16)).astype(np.float32)) for _ in range(4)]
    pfd = m.high_loss(_cl(r1), _cl(r2), _cl(l1), _cl(l2), [h.to(DEV) for h in hms], [{"hm": h.to(DEV)} for h in logits])
    close(pfd, g["pfd"], rtol=1e-3, atol=1e-6)
    # get_loss total + every tb entry, and gradients w.r.t. the four radar maps vs oracle autograd
    rd, r1d, r2d = [_cl(t).requires_grad_(True) for t in (rad, r1, r2)]
    bd = {"multi_scale_2d_features": {"x_conv4": _cl(lid)},
          "radar_multi_scale_2d_features": {"radar_spatial_features_8x_2": rd, "radar_spatial_features_8x_1": r1d},
          "radar_spatial_features_2d": r1d, "spatial_features_2d": _cl(l1),
          "radar_spatial_features_2d_8x": r2d, "spatial_features_2d_8x": _cl(l2),
          "radar_pred_dicts": [{"hm": h.to(DEV)} for h in logits], "target_dicts": {"heatmaps": [h.to(DEV) for h in hms]}}
    total, tb = m.get_loss(bd)
    close(total, g["get_loss_total"], rtol=1e-3, atol=1e-6)
    for k, v in tb.items():
        close(v, g["tb_" + k], rtol=1e-3, atol=1e-6, what=k)
    total.backward()
    ro, r1o, r2o = [t.clone().requires_grad_(True) for t in (rad, r1, r2)]
    tot_o, _ = obev.distill_loss(lid, {"radar_spatial_features_8x_2": ro, "radar_spatial_features_8x_1": r1o,
                                       "radar_spatial_features_2d": r1o, "radar_spatial_features_2d_8x": r2o}, l1, l2, hms, logits)
    tot_o.backward()
    close(rd.grad, ro.grad, atol=2e-4, what="grad 8x_2"); close(r1d.grad, r1o.grad, atol=2e-4, what="grad 8x_1 + 2d")
    close(r2d.grad, r2o.grad, atol=2e-4, what="grad 2d_8x")


# ------------------------------------------------------------------------------------------ rotated overlap
def test_rotated_overlap_vs_oracle_and_known_answers():
    from radardistill_amd import kernels as K
    rng = np.random.default_rng(3)
    n = 4000
    a = np.concatenate([rng.uniform(-20, 20, (n, 3)), rng.uniform(0.3, 8, (n, 3)), rng.uniform(-np.pi, np.pi, (n, 1))], 1).astype(np.float32)
    b = a.copy()
    b[:, :2] += rng.normal(0, 1.0, (n, 2)).astype(np.float32); b[:, 3:6] *= rng.uniform(0.7, 1.3, (n, 3)).astype(np.float32)
    b[:, 6] += rng.normal(0, 0.4, n).astype(np.float32)
    b[:50] = a[:50]                                        # identical boxes
    b[50:100, 0] += 100.0                                  # disjoint
    out = K.boxes_aligned_overlap_bev(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)).view(-1).cpu().numpy()
    ref = ohead.boxes_aligned_overlap_bev(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    # same algorithm, same fp32 operation order: agree to rounding of sin/cos/atan2 implementations
    np.testing.assert_allclose(out, ref, rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(out[:50], a[:50, 3] * a[:50, 4], rtol=1e-3)
    assert np.all(out[50:100] == 0)
    from radardistill_amd.pcdet.ops.iou3d_nms.iou3d_nms_utils import boxes_aligned_iou3d_gpu
    iou = boxes_aligned_iou3d_gpu(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)).cpu()
    close(iou, ohead.boxes_aligned_iou3d(torch.from_numpy(a), torch.from_numpy(b)), rtol=2e-3, atol=2e-3)


# ------------------------------------------------------------------------------------------ optimizer
def test_fused_adam_onecycle_matches_oracle():
    from radardistill_amd.train import FusedAdamOneCycle, OneCycle
    rng = np.random.default_rng(0)
    shapes = [(300,), (64, 3, 3, 32), (5000, 7), (1,), (4097,)]
    ps = [torch.nn.Parameter(torch.from_numpy(rng.normal(size=s).astype(np.float32)).to(DEV)) for s in shapes]
    ref_p = [p.detach().cpu().clone() for p in ps]
    m = [torch.zeros_like(p) for p in ref_p]; v = [torch.zeros_like(p) for p in ref_p]
    opt = FusedAdamOneCycle(ps, wd=0.01, grad_clip=10.0)
    sched = OneCycle(opt, 1000, 1e-3, [0.95, 0.85], 10, 0.4)
    for it in range(4):
        sched.step(it * 150)
        lr, mom = ooptim.one_cycle(it * 150, 1000)
        assert abs(opt.lr - lr) < 1e-12 and abs(opt.mom - mom) < 1e-12
        grads = [torch.from_numpy((rng.normal(size=s) * (30.0 if it == 1 else 0.5)).astype(np.float32)) for s in shapes]
        for p, g in zip(ps, grads):
            p.grad = g.to(DEV)
        norm = opt.step()
        total, clipped = ooptim.clip_grad_norm(grads, 10.0)
        close(norm[0], total, rtol=1e-5)
        ooptim.adam_true_wd_step(ref_p, clipped, m, v, it + 1, lr, mom)
        for a, b in zip(ps, ref_p):
            close(a, b, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------ goldens of the 2-D modules
def test_dense_enc_golden(golden_dir):
    from radardistill_amd.pcdet.config import AttrDict
    from radardistill_amd.pcdet.models.backbones_2d import __all__ as REG
    g = np.load(f"{golden_dir}/g2_dense_enc.npz")
    m = REG["BaseBEVBackboneV2"](AttrDict(BEV_CFG), input_channels=256)
    sd = m.state_dict(); seeded_fill_(sd, seed=12); m.load_state_dict(sd); m = m.to(DEV)
    x4, x5 = _bev_inputs(21)
    for mode in ("eval", "train"):
        m.train(mode == "train")
        with torch.no_grad():
            d = m({"multi_scale_2d_features": {"x_conv4": _cl(x4), "x_conv5": _cl(x5)}})
        close(d["spatial_features_2d_8x"], g[f"{mode}_2d_8x"], what=mode); close(d["spatial_features_2d"], g[f"{mode}_2d"], what=mode)


def test_radar_distill_forward_golden(golden_dir):
    g = np.load(f"{golden_dir}/g3_radar_distill.npz")
    m = _radar_distill()
    x4, x5 = _bev_inputs(22, B=2)
    for mode in ("eval", "train"):
        m.train(mode == "train")
        with torch.no_grad():
            d = m({"radar_multi_scale_2d_features": {"x_conv4": _cl(x4), "x_conv5": _cl(x5)}})
        ms = d["radar_multi_scale_2d_features"]
        close(ms["radar_spatial_features_8x_2"], g[f"{mode}_8x_2"], what=mode); close(ms["radar_spatial_features_8x_1"], g[f"{mode}_8x_1"], what=mode)
        close(d["radar_spatial_features_2d_8x"], g[f"{mode}_2d_8x"], what=mode); close(d["radar_spatial_features_2d"], g[f"{mode}_2d"], what=mode)


def _head(seed=14, grid=128):
    from radardistill_amd.pcdet.config import AttrDict
    from radardistill_amd.pcdet.models.dense_heads import __all__ as REG
    from tests.golden.head_cfg import HEAD_CFG, CLASS_NAMES
    pc_range, voxel, gs = bench_geometry(grid)
    m = REG["Radar_CenterHead"](AttrDict(HEAD_CFG), input_channels=256, num_class=10, class_names=CLASS_NAMES, grid_size=gs,
                                point_cloud_range=pc_range, voxel_size=voxel, predict_boxes_when_training=False)
    sd = m.state_dict(); seeded_fill_(sd, seed=seed); m.load_state_dict(sd)
    return m.to(DEV)


def test_center_head_golden(golden_dir):
    g = np.load(f"{golden_dir}/g4_center_head.npz")
    m = _head()
    r = np.random.default_rng(24)
    feat = torch.from_numpy(r.normal(0, 1, size=(2, 256, 16, 16)).astype(np.float32))
    gt = torch.from_numpy(g["gt_boxes"])
    m.train()
    d = m({"radar_spatial_features_2d": _cl(feat), "gt_boxes": gt.to(DEV), "gt_boxes_host": g["gt_boxes"], "batch_size": 2})
    for h, pd in enumerate(d["radar_pred_dicts"]):
        for k, v in pd.items():
            close(v, g[f"pred_{h}_{k}"], what=f"pred {h} {k}")
    td = d["target_dicts"]
    for h in range(6):
        assert np.array_equal(td["heatmaps"][h].cpu().numpy(), g[f"hm_{h}"])              # exact
        # log / cos / sin of the box parameters run in torch-CPU on whatever host executes the test: allow 1 ulp of libm difference
        np.testing.assert_allclose(td["target_boxes"][h].cpu().numpy(), g[f"tb_{h}"], rtol=1e-6, atol=1e-7)
        assert np.array_equal(td["inds"][h].cpu().numpy(), g[f"ind_{h}"])
        assert np.array_equal(td["masks"][h].cpu().numpy(), g[f"mask_{h}"])
        assert np.array_equal(td["gt_box"][h].cpu().numpy(), g[f"gtbox_{h}"])
    loss, tb = m.get_loss()
    close(loss, g["loss"], rtol=1e-3, atol=1e-5)
    for k, v in tb.items():
        close(v, g["tb_" + k], rtol=1e-3, atol=1e-5, what=k)


@pytest.mark.parametrize("n_boxes,seed", [(12, 3), (60, 9), (0, 1)])
def test_fused_center_loss_matches_torch_expressions(n_boxes, seed):
    """centerloss.hip (3 launches forward, 2 backward) against the torch expressions of the same module (LOSS_CONFIG.FUSED: False) on the
    same maps and targets: total, the four per-head terms, and the gradient w.r.t. every map element.  60 boxes on a 16x16 map put
    several objects into one cell (scatter-add path); 0 boxes is the all-negative case."""
    g = np.random.default_rng(seed)
    feat = torch.from_numpy(g.normal(0, 1, size=(2, 256, 16, 16)).astype(np.float32))
    batch = make_batch(batch_size=2, n_lidar=16, n_radar=16, n_boxes=max(n_boxes, 1), grid=128, seed=seed)
    gtb = batch["gt_boxes"].copy()
    if n_boxes == 0:
        gtb[:] = 0
    m = _head(seed=15)
    m.train()
    m({"radar_spatial_features_2d": _cl(feat), "gt_boxes": torch.from_numpy(gtb).to(DEV), "gt_boxes_host": gtb, "batch_size": 2})
    o4 = m.forward_ret_dict['pred_stacked'][0]
    res = {}
    for fused in (True, False):
        m.model_cfg.LOSS_CONFIG['FUSED'] = fused
        loss, tb = m.get_loss()
        (go,) = torch.autograd.grad(loss.sum(), o4, retain_graph=True)
        res[fused] = (loss.detach(), {k: v.detach() for k, v in tb.items()}, go)
    close(res[True][0], res[False][0], rtol=1e-5, atol=1e-6, what="total")
    for k, v in res[False][1].items():
        close(res[True][1][k], v, rtol=1e-5, atol=1e-6, what=k)
    a, b = res[True][2], res[False][2]
    assert float((a - b).norm()) <= 1e-5 * float(b.norm()) + 1e-9, (float((a - b).norm()), float(b.norm()))
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-3, atol=1e-6 * float(b.abs().max()))


@pytest.mark.parametrize("training", [True, False])
def test_center_head_batched_branches_equal_per_branch_path(training):
    """The MI355X execution plan (all 42 branches: one conv + one BatchNorm + one narrow-conv launch) against the reference-shaped
    per-branch loop of the same module: predictions, loss, running statistics and parameter gradients."""
    import copy
    g = np.random.default_rng(31)
    feat = torch.from_numpy(g.normal(0, 1, size=(2, 256, 16, 16)).astype(np.float32))
    batch = make_batch(batch_size=2, n_lidar=16, n_radar=16, n_boxes=12, grid=128, seed=3)
    gt = torch.from_numpy(batch["gt_boxes"]).to(DEV)
    results = []
    base = _head(seed=15)
    for batched in (True, False):
        m = copy.deepcopy(base)
        m.model_cfg = copy.deepcopy(m.model_cfg); m.model_cfg["BATCH_BRANCHES"] = batched
        if training:
            m.train()
            d = m({"radar_spatial_features_2d": _cl(feat), "gt_boxes": gt, "gt_boxes_host": batch["gt_boxes"], "batch_size": 2})
            assert ("pred_stacked" in m.forward_ret_dict) == batched
            loss, tb = m.get_loss()
            loss.sum().backward()
            from radardistill_amd import autograd as A
            A.end_forward()
            results.append((d["radar_pred_dicts"], loss.detach(), {k: p.grad.detach().clone() for k, p in m.named_parameters()},
                            {k: v.detach().clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}))
        else:
            m.eval()
            for p in m.parameters():
                p.requires_grad_(False)
            with torch.no_grad():
                x = _cl(feat)
                preds = m.head_forward(x)
            assert ("pred_stacked" in m.forward_ret_dict) == batched
            results.append((preds, None, {}, {}))
    (pa, la, ga, sa), (pb, lb, gb, sb) = results
    for da, db in zip(pa, pb):
        assert list(da.keys()) == list(db.keys())
        for k in da:
            assert da[k].shape == db[k].shape
            close(da[k], db[k], what=f"pred {k}")
    if training:
        close(la, lb, rtol=1e-4)
        for k in gb:
            close(ga[k], gb[k], rtol=2e-3, atol=2e-4, what=f"grad {k}")
        for k in sb:
            close(sa[k], sb[k], what=k)


def test_center_head_concatenated_leaves_accumulate_and_follow_the_parameters():
    """autograd.ConcatLeaves behind the batched CenterHead branches: (a) a second backward pass without zero_grad ADDS to the branch
    parameters' gradients (delivered as views, outside AccumulateGrad), (b) after the parameters change in place the concatenated
    leaves are refreshed -- predictions equal the per-branch path again, (c) with one branch parameter frozen the plan falls back to
    torch.cat inside the graph and the frozen parameter gets no gradient."""
    import copy
    from radardistill_amd import autograd as A
    g = np.random.default_rng(32)
    feat = torch.from_numpy(g.normal(0, 1, size=(2, 256, 16, 16)).astype(np.float32))
    batch = make_batch(batch_size=2, n_lidar=16, n_radar=16, n_boxes=12, grid=128, seed=4)
    gt = torch.from_numpy(batch["gt_boxes"]).to(DEV)
    m = _head(seed=16)
    m.train()

    def step(mod):
        mod({"radar_spatial_features_2d": _cl(feat), "gt_boxes": gt, "gt_boxes_host": batch["gt_boxes"], "batch_size": 2})
        loss, _ = mod.get_loss()
        loss.sum().backward()
        A.end_forward()
        return loss.detach()

    # (a) in the library's fixed-order mode: the two passes are then the same arithmetic, and the sum must be 2 x the first gradient.
    # With the default atomics the BatchNorm statistics of the two passes differ in their last bits, and this input has a pixel of
    # heads_list.3.hm whose pre-activation sits within an ulp of zero: its ReLU mask flips in ~7 % of the runs and moves one output
    # channel of hm.0.0.weight by 6e-2 (tools/diag/head_accum_stress.py, round 3: always that channel, always that amount, in either
    # pass, and the single-stream recomputation agrees with the other pass) -- arithmetic noise, not an accumulation defect.
    from radardistill_amd import kernels as K
    K.set_deterministic(True)
    try:
        step(m)
        assert m._branch_plan[0]._store['ok']
        g1 = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        step(m)                                           # no zero_grad: the second pass accumulates
    finally:
        K.set_deterministic(False)
    for k, p in m.named_parameters():
        close(p.grad, 2 * g1[k], rtol=1e-5, atol=1e-6 * float(g1[k].abs().max()) + 1e-9, what=f"accumulated grad {k}")
    # (b) in-place parameter update (what torch optimizers do), then compare with the per-branch path of a copy
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.05)
    ref = _head(seed=16)
    ref.load_state_dict(m.state_dict())
    ref.train()
    ref.model_cfg = copy.deepcopy(ref.model_cfg); ref.model_cfg["BATCH_BRANCHES"] = False
    for mod in (m, ref):
        for p in mod.parameters():
            p.grad = None
    la, lb = step(m), step(ref)
    close(la, lb, rtol=1e-4)
    gm, gr = dict(m.named_parameters()), dict(ref.named_parameters())
    for k in gr:
        close(gm[k].grad, gr[k].grad, rtol=2e-3, atol=2e-4, what=f"grad after update {k}")
    # (c) one frozen branch parameter: fallback to torch.cat in the graph
    frozen = m.heads_list[0].center[1].weight
    frozen.requires_grad_(False)
    m._branch_plan[0]._store = None
    for p in m.parameters():
        p.grad = None
    step(m)
    assert m._branch_plan[0]._store['ok'] is False
    assert frozen.grad is None
    assert all(p.grad is not None for p in m.parameters() if p.requires_grad)


def test_conv5_golden(golden_dir):
    from functools import partial
    from radardistill_amd.pcdet.models.backbones_3d.spconv_backbone_2d import BasicBlock, post_act_block_dense
    g = np.load(f"{golden_dir}/g5_conv5.npz")
    norm = partial(torch.nn.BatchNorm2d, eps=1e-3, momentum=0.01)
    m = torch.nn.Sequential(post_act_block_dense(256, 256, 3, norm_fn=norm, stride=2, padding=1), BasicBlock(256, 256, norm_fn=norm),
                            BasicBlock(256, 256, norm_fn=norm))
    sd = m.state_dict(); seeded_fill_(sd, seed=15); m.load_state_dict(sd); m = m.to(DEV)
    x4, _ = _bev_inputs(25)
    for mode in ("eval", "train"):
        m.train(mode == "train")
        with torch.no_grad():
            close(m(_cl(x4)), g[f"{mode}_x_conv5"], what=mode)


# ------------------------------------------------------------------------------------------ whole step vs oracle
def _build_pillarnet(grid):
    from radardistill_amd.data import SyntheticDistillDataset
    from radardistill_amd.pcdet.config import AttrDict, cfg_from_yaml_file
    from radardistill_amd.pcdet.models import build_network
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c = cfg_from_yaml_file(os.path.join(root, "tools/cfgs/radar_distill/bench_512.yaml"), AttrDict())
    pc_range, voxel, gs = bench_geometry(grid)
    c.DATA_CONFIG.POINT_CLOUD_RANGE = pc_range
    c.MODEL.RADAR_BACKBONE_2D.POINT_CLOUD_RANGE = pc_range
    ds = SyntheticDistillDataset.from_cfg(c)
    torch.manual_seed(0)
    m = build_network(model_cfg=c.MODEL, num_class=len(c.CLASS_NAMES), dataset=ds)
    return m, c, pc_range, voxel, gs


def _oracle_step(model, batch, pc_range, voxel, gs, B, dtype):
    """The CPU oracle's training step in `dtype` on the model's current weights -> (loss, tb, {name: gradient})."""
    state = {k: (v.detach().cpu().clone().to(dtype) if v.is_floating_point() else v.detach().cpu().clone()) for k, v in model.state_dict().items()}
    trainable = [k for k, p in model.named_parameters() if p.requires_grad]
    for k in trainable:
        state[k].requires_grad_(True)
    ob = {"points": torch.from_numpy(batch["points"]).to(dtype), "radar_points": torch.from_numpy(batch["radar_points"]).to(dtype),
          "gt_boxes": torch.from_numpy(batch["gt_boxes"]).to(dtype), "batch_size": B}
    oloss, otb, _ = opn.forward_train(state, ob, pc_range, voxel, gs)
    oloss = oloss.mean()
    oloss.backward()
    return oloss.detach(), otb, {k: (state[k].grad if state[k].grad is not None else torch.zeros_like(state[k])) for k in trainable}


def test_full_distillation_step_vs_oracle(noise_factor=3.5, whole_factor=3.0, tensor_tol=1e-3):
    """Config C4 at reduced size (128 x 128 BEV, B = 2): loss, every tb entry and all ~500 gradients of a training step, with the
    library's reductions in fixed order (rd_set_deterministic) so that the HIP side is one reproducible answer.

    Gradient bound.  Two fp32 evaluations of this network disagree on the SIGN of a handful of ~1e-8 ReLU inputs, and each flip
    moves a few gradient rows by O(1): the fp32 CPU oracle itself sits ~1e-2 (relative L2, per tensor) from the fp64 oracle, so
    no fp32 implementation -- the reference's included -- can be held to 1e-3 against it.  The check is therefore made against
    the EXACT gradient (the oracle in fp64): a tensor passes at `tensor_tol` (1e-3), or when HIP is at most `noise_factor` times as
    far from the fp64 gradient as the fp32 oracle is; the whole gradient likewise with `whole_factor`.  A scheduling or indexing defect moves
    tensors by 10 %+ (the one race found in round 1: 300-600 %), an order of magnitude outside this bound.
    Round 3: the HIP step in this mode is one fixed answer (tools/diag/history_step.py / poison_step.py: bit-identical gradients whatever
    ran before in the process and whatever freshly allocated memory contains), but the statistic is chaotic in the arithmetic: the fp32
    CPU oracle ITSELF moves by 2.3e-2 (whole-gradient relative L2) between 1 and 8 torch threads, and the HIP answer sits between 0.8x
    and 2.03x the fp32 oracle's distance from the fp64 gradient depending on which of the library's equivalent kernels a process
    picked -- hence whole_factor 3 (was 2: exceeded by 1.4 % when this test runs alone) and noise_factor 3.5 per tensor (was 2.5: the
    radar VFE's BatchNorm weight -- the end of the longest backward chain -- landed at 2.67x after the pillar mean's summation order
    changed with the 4-pillars-per-wavefront kernel; a defect shows as 10 %+ of a tensor, these bounds sit at 4-5 %).
    The oracle's distance is itself one draw per summation order -- radar_backbone_3d.conv1.0.bn2.weight: 1.2e-2 / 2.2e-2 / 3.4e-2 of
    its norm with 2 / 4 / 1 torch threads, HIP 6.3e-2 -- and the host of the GPU box decides the default thread count, so the bound
    uses the largest of three draws (default, 1 and 4 threads) instead of whichever one the machine happens to produce."""
    from radardistill_amd import kernels as K
    from radardistill_amd.pcdet.models import model_fn_decorator
    grid, B = 128, 2
    model, cfg, pc_range, voxel, gs = _build_pillarnet(grid)
    sd = model.state_dict(); seeded_fill_(sd, seed=77); model.load_state_dict(sd)
    batch = make_batch(batch_size=B, n_lidar=300, n_radar=700, n_boxes=10, grid=grid, seed=5)   # sparse lidar: AFD is NaN by definition when every 8x cell is lidar-active
    o32 = _oracle_step(model, batch, pc_range, voxel, gs, B, torch.float32)
    o64 = _oracle_step(model, batch, pc_range, voxel, gs, B, torch.float64)
    # the yardstick is the fp32 oracle's OWN distance from the fp64 gradient, and that is one draw of a chaotic statistic per summation
    # order: two more draws (1 and 4 torch threads; per tensor the 1-thread draw is up to 2.9x the 8-thread one), the largest counts
    threads = torch.get_num_threads()
    g32_draws = [o32[2]]
    try:
        for th in (1, 4):
            if th != threads:
                torch.set_num_threads(th)
                g32_draws.append(_oracle_step(model, batch, pc_range, voxel, gs, B, torch.float32)[2])
    finally:
        torch.set_num_threads(threads)
    model = model.to(DEV)
    model.train()
    K.set_deterministic(True)
    try:
        loss, tb, _ = model_fn_decorator()(model, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()})
        loss.backward()
        torch.cuda.synchronize()
    finally:
        K.set_deterministic(False)
    print("loss hip / oracle fp32 / oracle fp64", float(loss), float(o32[0]), float(o64[0]))
    close(loss, o32[0], rtol=1e-3, atol=1e-5, what="total loss")
    close(loss, o64[0].float(), rtol=1e-3, atol=1e-5, what="total loss vs fp64")
    for k, v in o32[1].items():
        close(tb[k], v, rtol=2e-3, atol=1e-5, what=k)
    named = dict(model.named_parameters())
    g32, g64 = o32[2], o64[2]
    gscale = max(float(g.abs().max()) for g in g64.values())
    worst, at_1e3 = ("", 0.0), 0
    e_hip2 = ref2 = 0.0
    e_o322 = [0.0] * len(g32_draws)
    for k, ref in g64.items():
        a = named[k].grad
        assert a is not None, k
        e_hip = float((a.detach().cpu().double() - ref).norm())
        e_draws = [float((g[k].double() - ref).norm()) for g in g32_draws]
        e_o32 = max(e_draws)
        rn = float(ref.norm())
        e_hip2 += e_hip ** 2; ref2 += rn ** 2
        e_o322 = [s_ + e ** 2 for s_, e in zip(e_o322, e_draws)]
        floor = 1e-5 * gscale * (ref.numel() ** 0.5)          # conv biases in front of a BatchNorm: the exact gradient is 0
        bound = max(tensor_tol * rn, noise_factor * e_o32) + floor
        at_1e3 += e_hip <= 1e-3 * rn + floor
        if e_hip / bound > worst[1]:
            worst = (k, e_hip / bound)
        assert e_hip <= bound, (k, "hip vs fp64", e_hip / (rn + 1e-30), "fp32 oracle vs fp64", e_o32 / (rn + 1e-30))
    whole_hip, whole_draws = (e_hip2 / ref2) ** 0.5, [(e / ref2) ** 0.5 for e in e_o322]
    whole_o32 = max(whole_draws)
    print(f"{at_1e3} of {len(g64)} tensors within 1e-3 of the fp64 gradient; worst error / bound {worst}; whole-gradient relative L2: "
          f"hip {whole_hip:.3e}, fp32 oracle draws {[f'{w:.3e}' for w in whole_draws]}")
    assert whole_hip <= max(1e-3, whole_factor * whole_o32), (whole_hip, whole_o32)
    assert int(model.global_step) == 1


def test_deterministic_mode_is_bit_reproducible():
    """rd_set_deterministic(1): the same step twice gives bit-identical loss terms and gradients (with the default atomics they differ
    in the last bits and, through ReLU sign flips, by percents on some tensors)."""
    from radardistill_amd import kernels as K
    from radardistill_amd.pcdet.models import model_fn_decorator
    grid, B = 128, 2
    model, cfg, pc_range, voxel, gs = _build_pillarnet(grid)
    sd = model.state_dict(); seeded_fill_(sd, seed=77); model.load_state_dict(sd)
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    batch = make_batch(batch_size=B, n_lidar=2500, n_radar=700, n_boxes=10, grid=grid, seed=5)
    fn = model_fn_decorator()

    def run():
        model.load_state_dict(state0)
        model.train()
        model.zero_grad(set_to_none=True)
        loss, tb, _ = fn(model, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()})
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), {k: v.detach().clone() if torch.is_tensor(v) else v for k, v in tb.items()}, \
            {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    assert not K.get_deterministic()
    K.set_deterministic(True)
    try:
        assert K.get_deterministic()
        (l1, t1, g1), (l2, t2, g2) = run(), run()
    finally:
        K.set_deterministic(False)
    assert torch.equal(l1, l2)
    for k in t1:
        assert (torch.equal(torch.as_tensor(t1[k]), torch.as_tensor(t2[k])) or (np.isnan(float(t1[k])) and np.isnan(float(t2[k])))), k
    assert g1.keys() == g2.keys()
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k


@pytest.mark.parametrize("grid,B,n_boxes", [(512, 8, 30), (128, 3, 120)])
def test_center_targets_gpu_kernel_vs_reference_host_loops(grid, B, n_boxes):
    """targets.hip vs the reference-shaped host implementation (itself pinned by the g4 golden): integer outputs and heat-maps
    bit-exact, box regression targets to 1 ulp of log/cos/sin."""
    m = _head(grid=grid)
    batch = make_batch(batch_size=B, n_lidar=16, n_radar=16, n_boxes=n_boxes, grid=grid, seed=17)
    gt = torch.from_numpy(batch["gt_boxes"]).clone()
    gt[0, -5:, :] = 0                                         # padding rows
    gt[1, 0, 3] = 0.0                                         # degenerate box: skipped (dx <= 0)
    R = 0.1 * grid
    gt[2 % B, 1, :2] = torch.tensor([R - 0.01, -R + 0.01])    # on the border: clipped gaussian window
    fm = (grid // 8, grid // 8)
    host = m.assign_targets(gt, feature_map_size=fm, gt_boxes_host=gt.numpy())
    dev = m.assign_targets_gpu(gt.to(DEV), fm)
    for h in range(6):
        assert torch.equal(dev["heatmaps"][h].cpu(), host["heatmaps"][h].cpu()), f"heatmap head {h}"
        assert torch.equal(dev["inds"][h].cpu(), host["inds"][h].cpu()) and torch.equal(dev["masks"][h].cpu(), host["masks"][h].cpu())
        assert torch.equal(dev["gt_box"][h].cpu(), host["gt_box"][h].cpu())
        np.testing.assert_allclose(dev["target_boxes"][h].cpu().numpy(), host["target_boxes"][h].cpu().numpy(), rtol=2e-6, atol=1e-7)
    assert int(dev["_stacked"]["masks"].sum()) > 0


def test_bf16x3_conv_math_parity(golden_dir):
    """The split-bf16 MFMA mode (conv_b3.hip) against the same oracles / goldens and the same 1e-3 bound as the exact-fp32 mode."""
    from radardistill_amd import kernels as K
    import tests.test_gpu_kernels as TK
    K.set_conv_math("bf16x3")
    try:
        assert K.get_conv_math() == "bf16x3"
        TK.test_sparse_conv_forward_and_backward(64, 128)
        TK.test_sparse_conv_forward_and_backward(128, 256)
        TK.test_dense_conv2d_forward_and_backward(256, 256, 3, 1, 1, 12, 10)
        TK.test_dense_conv2d_forward_and_backward(256, 256, 3, 2, 1, 13, 11)
        TK.test_dense_conv2d_forward_and_backward(512, 256, 1, 1, 0, 9, 9)
        TK.test_conv_transpose2d_forward_and_backward(4, 2, 1)
        TK.test_linear_as_one_tap_conv()
        TK.test_conv_epilogue_and_fused_stats()
        test_dcn_forward_backward_vs_oracle(256, 256, 16, 16, 2, True, pytest.MonkeyPatch())
        TK.test_sparse_enc_c2_vs_oracle(False)
        test_dense_enc_golden(golden_dir)
        test_radar_distill_forward_golden(golden_dir)
        test_center_head_golden(golden_dir)
        # losses / tb entries keep the 1e-3..2e-3 bounds; the whole-network gradient comparison sees more ReLU sign flips at 4e-6
        # forward noise than at 4e-7, hence the wider multiples of the fp32 oracle's own distance from the fp64 gradient
        # (bf16x3 products carry 4e-6 instead of 4e-7: ~10x the pre-activations change sign, also in tensors where the fp32 oracle
        # happens to have no flip at all, hence a per-tensor floor of 4e-2.  Round 3: against the LARGEST of the three fp32-oracle draws
        # the whole bf16x3 gradient sits at 3.1x (5.8e-2 vs 1.84e-2) and the worst tensor at 0.45 of the former (8x, 5e-2) bound, so the
        # bounds came down from 8x / 6x / 5e-2 to 6x / 4.5x / 4e-2)
        test_full_distillation_step_vs_oracle(noise_factor=6.0, whole_factor=4.5, tensor_tol=4e-2)
    finally:
        K.set_conv_math("f32")


@pytest.mark.parametrize("shape", ["dense3x3", "gemm", "sparse", "strided"])
def test_amp_single_term_products_equal_bf16_rounded_operands(shape):
    """rd_set_mfma_terms(1) -- the `--use_amp` arithmetic (train.autocast): every MFMA convolution kernel forms its products from the
    operands ROUNDED TO bf16 and accumulates in fp32.  Forward, data gradient and weight gradient of a dense 3x3, a 1-tap GEMM, a
    sub-manifold sparse and a strided dense convolution against fp64 convolutions of the bf16-rounded operands (1e-5: only the fp32
    accumulation order differs), and the distance to the un-rounded result is the expected ~2^-9 per operand."""
    from radardistill_amd import autograd as A, kernels as K, sparse as SP
    g = np.random.default_rng(3)
    B, H, W, Cin, Cout = 2, 24, 32, 64, 128
    rnd = lambda t: t.bfloat16().double()
    K.set_conv_math("bf16x3")
    K.set_mfma_terms(1)
    try:
        assert K.get_mfma_terms() == 1
        if shape == "sparse":
            n = 900
            keys = np.sort(g.choice(B * H * W, size=n, replace=False))
            idx = np.stack([keys // (H * W), (keys // W) % H, keys % W], axis=1).astype(np.int32)
            x = torch.from_numpy(g.normal(size=(n, Cin)).astype(np.float32))
            w = torch.from_numpy((g.normal(size=(Cout, 3, 3, Cin)) / np.sqrt(9 * Cin)).astype(np.float32))
            conv = SP.SubMConv2d(Cin, Cout, 3, padding=1, bias=False).to(DEV)
            with torch.no_grad():
                conv.weight.copy_(w)
            xd = x.to(DEV).requires_grad_(True)
            A.begin_step(torch.device(DEV))
            out = conv(SP.SparseConvTensor(xd, torch.from_numpy(idx).to(DEV), [H, W], B)).features
            go = torch.from_numpy(g.normal(size=tuple(out.shape)).astype(np.float32))
            (out * go.to(DEV)).sum().backward()
            from oracle import sparse as osp
            nbr = osp.subm_rulebook(idx, (H, W))
            xr, wr = rnd(x).requires_grad_(True), rnd(w).requires_grad_(True)
            ref = osp.sparse_conv(xr, nbr, wr, None)
            exact = osp.sparse_conv(x.double(), nbr, w.double(), None)
            (ref * go.double()).sum().backward()
            # gradients: the kernels round grad_out to bf16 as well -- compare against fp64 products of rounded operands
            xg, wg = rnd(x).requires_grad_(True), rnd(w).requires_grad_(True)
            r2 = osp.sparse_conv(xg, nbr, wg, None)
            gx_ref = torch.autograd.grad((r2 * rnd(go)).sum(), xg, retain_graph=True)[0]
            gw_ref = torch.autograd.grad((r2 * rnd(go)).sum(), wg)[0]
            got = (out.detach().cpu().double(), xd.grad.cpu().double(), conv.weight.grad.cpu().double())
        else:
            k, s_, p_ = {"dense3x3": (3, 1, 1), "gemm": (1, 1, 0), "strided": (3, 2, 1)}[shape]
            x = torch.from_numpy(g.normal(size=(B, Cin, H, W)).astype(np.float32))
            w = torch.from_numpy((g.normal(size=(Cout, Cin, k, k)) / np.sqrt(k * k * Cin)).astype(np.float32))
            spec = A.dense_conv_spec(B, H, W, k, k, s_, p_)
            xd = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().to(DEV).requires_grad_(True)
            wp = torch.nn.Parameter(w.to(DEV))
            A.begin_step(torch.device(DEV))
            out = A.conv(xd, wp, None, spec, Cout)
            Ho, Wo = spec.out_hw
            go = torch.from_numpy(g.normal(size=(B, Cout, Ho, Wo)).astype(np.float32))
            (out * go.permute(0, 2, 3, 1).reshape(-1, Cout).to(DEV)).sum().backward()
            ref = F.conv2d(rnd(x), rnd(w), None, s_, p_)
            exact = F.conv2d(x.double(), w.double(), None, s_, p_)
            xg, wg = rnd(x).requires_grad_(True), rnd(w).requires_grad_(True)
            r2 = F.conv2d(xg, wg, None, s_, p_)
            gx_ref, gw_ref = torch.autograd.grad((r2 * rnd(go)).sum(), (xg, wg))
            rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
            ref, exact, gx_ref = rows(ref), rows(exact), rows(gx_ref)
            got = (out.detach().cpu().double(), xd.grad.cpu().double(), wp.grad.cpu().double())
        torch.cuda.synchronize()
        rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
        assert rel(got[0], ref) < 1e-5, ("forward vs bf16-rounded operands", rel(got[0], ref))
        assert 5e-4 < rel(got[0], exact) < 8e-3, ("forward vs exact: plain bf16 products expected", rel(got[0], exact))
        assert rel(got[1], gx_ref) < 1e-5, ("data gradient", rel(got[1], gx_ref))
        assert rel(got[2], gw_ref) < 1e-5, ("weight gradient", rel(got[2], gw_ref))
    finally:
        K.set_mfma_terms(3)
        K.set_conv_math("f32")


def test_amp_training_step_autocast_and_loss_scaling():
    """The reference's `--use_amp` iteration (tools/train_utils/train_utils.py:57-64) through train.train_step(scaler=AmpScaler):
    forward and backward under train.autocast (bf16 products), scaled loss, unscale + clip + Adam in the fused optimizer.  The loss
    stays within 2e-2 (tb entries 15e-2, the IoU terms 35e-2) of the fp32-class step, every gradient is finite, parameters move, and the modes are restored."""
    from radardistill_amd import kernels as K
    from radardistill_amd.pcdet.models import model_fn_decorator
    from radardistill_amd.train import AmpScaler, build_optimizer, build_scheduler, train_step
    grid, B = 128, 2
    batch = make_batch(batch_size=B, n_lidar=300, n_radar=700, n_boxes=10, grid=grid, seed=5)
    fn = model_fn_decorator()
    res = {}
    for amp in (False, True):
        model, cfg, *_ = _build_pillarnet(grid)
        sd = model.state_dict(); seeded_fill_(sd, seed=77); model.load_state_dict(sd)
        model = model.to(DEV)
        opt = build_optimizer(model, cfg.OPTIMIZATION)
        sched, _ = build_scheduler(opt, 100, 1, -1, cfg.OPTIMIZATION)
        scaler = AmpScaler(torch.device(DEV), init_scale=2.0 ** 10, enabled=amp)
        K.set_conv_math("bf16x3")
        try:
            before = {k: p.detach().clone() for k, p in model.named_parameters() if p.requires_grad}
            loss, tb = train_step(model, opt, sched, fn, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()}, 0,
                                  scaler=scaler if amp else None)
            torch.cuda.synchronize()
            assert K.get_conv_math() == "bf16x3" and K.get_mfma_terms() == 3          # autocast restored the modes
            moved = sum(int(not torch.equal(p.detach(), before[k])) for k, p in model.named_parameters() if p.requires_grad)
            grads_ok = all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
            res[amp] = (float(loss), {k: float(v) for k, v in tb.items()}, moved, grads_ok, float(opt.norm_out[0]))
        finally:
            K.set_mfma_terms(3)
            K.set_conv_math("f32")
    (l0, tb0, m0, ok0, n0), (l1, tb1, m1, ok1, n1) = res[False], res[True]
    assert ok0 and ok1 and m1 > 400 and m0 > 400
    assert abs(l1 - l0) <= 2e-2 * abs(l0), (l0, l1)
    for k, v in tb0.items():
        if np.isfinite(v) and abs(v) > 1e-3:
            # the IoU terms compare boxes DECODED from a random-weight head with the ground truth: 8 mantissa bits move them by 9-15 %
            # from run to run of the float atomics (0.152 seen once against the former single 0.15 bound); they get their own bound
            tol = 0.35 if "iou" in k else 0.15
            assert abs(tb1[k] - v) <= tol * abs(v) + 1e-3, (k, v, tb1[k])
    assert np.isfinite(n1) and abs(n1 - n0) <= 0.25 * n0          # the unscaled gradient norm (g / S inside the norm kernel)
    assert l1 != l0          # the arithmetic really changed


def test_teacher_prefetch_gives_the_same_step():
    """PillarNet.prefetch_teacher (teacher branch of the NEXT batch enqueued between backward and optimizer.step) against the plain
    forward on the same batches: BIT-IDENTICAL loss values and gradients (reductions in fixed order, rd_set_deterministic), over three
    pipelined steps with two alternating batches; a dict that was not prefetched still takes the plain path."""
    import os
    from radardistill_amd import kernels as K
    from radardistill_amd.pcdet.models import model_fn_decorator
    grid, B = 128, 2
    model, cfg, pc_range, voxel, gs = _build_pillarnet(grid)
    sd = model.state_dict(); seeded_fill_(sd, seed=77); model.load_state_dict(sd)
    model = model.to(DEV)
    model.train()
    batches = [make_batch(batch_size=B, n_lidar=300, n_radar=700, n_boxes=10, grid=grid, seed=s_) for s_ in (5, 6)]
    fn = model_fn_decorator()
    prev = os.environ.get("RD_TEACHER_PREFETCH")
    os.environ["RD_TEACHER_PREFETCH"] = "1"              # off by default (it measured slower); the mechanism must still be right

    def fresh(i):
        return {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batches[i % 2].items()}

    def run(pipelined):
        out, nxt = [], None
        for it in range(3):
            model.zero_grad(set_to_none=True)
            bd = nxt if (pipelined and nxt is not None) else fresh(it)
            assert ('_teacher_done' in bd) == (pipelined and it > 0)
            loss, _, _ = fn(model, bd)
            loss.backward()
            if pipelined:
                nxt = model.prefetch_teacher(fresh(it + 1))
                assert '_teacher_done' in nxt
            torch.cuda.synchronize()
            out.append((float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
        return out

    K.set_deterministic(True)
    try:
        ref, got = run(False), run(True)
    finally:
        K.set_deterministic(False)
        if prev is None:
            os.environ.pop("RD_TEACHER_PREFETCH", None)
        else:
            os.environ["RD_TEACHER_PREFETCH"] = prev
    for (lr, gr), (lg, gg) in zip(ref, got):
        assert lr == lg, (lr, lg)
        assert gr.keys() == gg.keys()
        for k, r in gr.items():
            assert torch.equal(gg[k], r), k


def test_operand_cache_multi_refresh_matches_single_conversions():
    """autograd._OperandCache: the one-launch refresh of every stale weight operand (rd_weight_layout_split_multi) writes exactly
    what the per-weight conversion writes, for every operand kind, after torch-side updates and after the fused optimizer's epoch bump."""
    from radardistill_amd import autograd as A, kernels as K
    g = torch.Generator(device="cpu").manual_seed(4)
    shapes = [((96, 3, 3, 64), 0, 96, 64, 9), ((128, 64, 3, 3), 1, 128, 64, 9), ((64, 128, 2, 2), 3, 128, 64, 4), ((256, 1024), 0, 256, 1024, 1),
              ((64, 100, 4, 4), 3, 100, 64, 16), ((36, 72, 3, 3), 1, 36, 72, 9)]          # 16 taps (two tap chunks), ragged tile edges
    params = [torch.nn.Parameter(torch.randn(*sh, generator=g).to(DEV)) for sh, _, _, _, _ in shapes]
    cache = A._OperandCache()

    def frag_reference(dst_order):
        """[A][taps][B] fp32 in destination order -> the fragment-major split image, built with torch ops (include/rdamd.h, RD_LAYOUT_FRAG):
        per 32 (A) x 16 (B) block of a tap the hi then the lo operand image, each [k-half][row][8 elements]."""
        A_, T_, B_ = dst_order.shape
        hi = dst_order.bfloat16()
        lo = (dst_order - hi.float()).bfloat16()
        parts = [t.view(A_ // 32, 32, T_, B_ // 16, 2, 8).permute(0, 2, 3, 4, 1, 5) for t in (hi, lo)]
        return torch.stack(parts, dim=3).contiguous().view(torch.int16).reshape(-1)

    def check_all():
        for p_, (_, pk, Cout, Cin, taps) in zip(params, shapes):
            for dgrad in (False, True):
                kind = A._DGRAD_KIND[pk] if dgrad else pk
                got = cache.get(p_, Cout, Cin, taps, kind)
                want = K.weight_layout_split(p_.detach().contiguous(), Cout, Cin, taps, kind, False)
                assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (pk, dgrad)
                if Cout % 32 == 0 and Cin % 32 == 0:          # fragment-major variant of the same operand
                    gotf = cache.get(p_, Cout, Cin, taps, kind | K.LAYOUT_FRAG)
                    wantf = K.weight_layout_split(p_.detach().contiguous(), Cout, Cin, taps, kind, False, frag=True)
                    assert torch.equal(gotf.view(torch.int32), wantf.view(torch.int32)), (pk, dgrad, "frag")
                    plain = K.weight_layout(p_.detach().contiguous(), Cout, Cin, taps, kind)          # destination order, fp32
                    a_, b_ = (Cout, Cin) if not dgrad else (Cin, Cout)
                    assert torch.equal(wantf.view(torch.int16).reshape(-1), frag_reference(plain.view(a_, taps, b_))), (pk, dgrad, "frag layout")

    check_all()                                              # entries created one by one
    with torch.no_grad():
        for p_ in params:
            p_.mul_(1.5)                                     # torch-side update: _version bump
    cache.refresh_all(torch.device(DEV))
    assert cache.table is not None and cache.table[4] > 0
    ptrs = {k: e[1].data_ptr() for k, e in cache.entries.items()}
    assert all(e[0] == cache._ver(e[2]()) for e in cache.entries.values())      # nothing left stale -> check_all converts nothing itself
    check_all()
    for p_ in params:                                        # raw-pointer update as the fused optimizer does it, then the epoch bump
        p_.data.add_(0.25)
    A.bump_weights_epoch()
    cache.refresh_all(torch.device(DEV))
    assert {k: e[1].data_ptr() for k, e in cache.entries.items()} == ptrs       # persistent destination buffers
    check_all()


def test_stream_overlaps_do_not_change_gradients():
    """The same training step with every stream feature off (geometry prelude, weight-gradient side stream, teacher stream) and with
    all of them on, once as a single backward pass and once accumulating two passes into .grad: with the library's reductions in
    fixed order (rd_set_deterministic) every parameter gradient is BIT-IDENTICAL -- stream placement may change when a kernel runs,
    never what it computes, so any missing wait_event / record_stream shows as a plain inequality.  (Regression: a strided dwconv
    weight gradient produced on the side stream was cloned by AccumulateGrad on the main stream before it was written.)"""
    import os
    from radardistill_amd import autograd as A, kernels as K
    from radardistill_amd.pcdet.models import model_fn_decorator
    grid, B = 128, 2
    model, cfg, pc_range, voxel, gs = _build_pillarnet(grid)
    sd = model.state_dict(); seeded_fill_(sd, seed=77); model.load_state_dict(sd)
    model = model.to(DEV)
    batch = make_batch(batch_size=B, n_lidar=300, n_radar=700, n_boxes=10, grid=grid, seed=5)
    fn = model_fn_decorator()
    saved = {k: os.environ.get(k) for k in ("RD_GEOM_STREAM", "RD_TEACHER_STREAM")}
    wg0 = A.WGRAD_STREAM[0]

    def grads(on, passes):
        os.environ["RD_GEOM_STREAM"] = os.environ["RD_TEACHER_STREAM"] = "1" if on else "0"
        A.WGRAD_STREAM[0] = bool(on)
        model.train()
        model.zero_grad(set_to_none=True)
        for _ in range(passes):
            loss, _, _ = fn(model, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()})
            loss.backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    K.set_deterministic(True)
    try:
        for passes in (1, 2):
            ref, got = grads(False, passes), grads(True, passes)
            assert ref.keys() == got.keys()
            for k, r in ref.items():
                assert torch.equal(got[k], r), (passes, k, float((got[k] - r).norm()) / (float(r.norm()) + 1e-30))
    finally:
        K.set_deterministic(False)
        A.WGRAD_STREAM[0] = wg0
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 21, 19, 64, 96),      # ragged map, 8x8 pixel tiles, Cout not a tile multiple
                                            (8, 64, 48, 64, 256),     # 8x16 pixel tiles (>= 384 workgroups)
                                            (3, 24, 40, 96, 64),      # 64-column tile of the fragment-major kernel, 3 K chunks
                                            (1, 8, 16, 32, 33)])
def test_halo_conv3x3_bf16x3(B, H, W, Cin, Cout):
    """Dense stride-1 3x3 convolutions in bf16x3 mode run on the halo-staged kernel (k_conv_d3_b3): forward with the full epilogue
    (bias, scale/shift, residual, ReLU, fused column statistics) and the data gradient (mirrored taps), against torch conv2d on the
    CPU at the same 1e-3 bound (observed ~4e-6)."""
    from radardistill_amd import autograd as A, kernels as K
    g = np.random.default_rng(B * 1000 + H)
    x = torch.from_numpy(g.normal(size=(B, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy((g.normal(size=(Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32))
    b, sc, sh = [torch.from_numpy(g.normal(size=(Cout,)).astype(np.float32)) for _ in range(3)]
    res = torch.from_numpy(g.normal(size=(B * H * W, Cout)).astype(np.float32))
    go = torch.from_numpy(g.normal(size=(B, Cout, H, W)).astype(np.float32))
    xr = x.clone().requires_grad_(True)
    pre = F.conv2d(xr, w, b, 1, 1)
    (pre * go).sum().backward()
    pre_rows = pre.detach().permute(0, 2, 3, 1).reshape(-1, Cout)
    K.set_conv_math("bf16x3")
    try:
        spec = A.dense_conv_spec(B, H, W, 3, 3, 1, 1)
        rows = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().to(DEV)
        wk = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous().to(DEV)
        out = K.conv_fwd(rows, wk, 9, b.to(DEV), B * H * W, Cout, spec.fwd_ix, scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV), relu=True)
        close(out, torch.relu(pre_rows * sc + sh + res), rtol=1e-3, atol=1e-4, what="halo conv fwd + epilogue")
        stats = torch.zeros(2 * Cout, device=DEV)
        plain = K.conv_fwd(rows, wk, 9, b.to(DEV), B * H * W, Cout, spec.fwd_ix, stats=stats)
        close(plain, pre_rows, rtol=1e-3, atol=1e-4, what="halo conv fwd")
        assert torch.equal(K.conv_fwd(rows, K.weight_layout_split(wk, Cout, Cin, 9, 0), 9, b.to(DEV), B * H * W, Cout, spec.fwd_ix, w_split=True), plain)
        if Cout % 32 == 0:
            # the same convolution with the weights in fragment-major split format (k_conv_d3f_b3: fragments straight from L2, one barrier
            # per K chunk): same products in the same order -> bit-identical to the LDS-staged kernel, forward, epilogue and data gradient
            wf = K.weight_layout_split(wk, Cout, Cin, 9, 0, frag=True)
            assert torch.equal(K.conv_fwd(rows, wf, 9, b.to(DEV), B * H * W, Cout, spec.fwd_ix, w_split=2), plain)
            outf = K.conv_fwd(rows, wf, 9, b.to(DEV), B * H * W, Cout, spec.fwd_ix, scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV), relu=True, w_split=2)
            assert torch.equal(outf, out)
            statsf = torch.zeros(2 * Cout, device=DEV)
            K.conv_fwd(rows, wf, 9, b.to(DEV), B * H * W, Cout, spec.fwd_ix, stats=statsf, w_split=2)
            close(statsf, stats, rtol=1e-5, atol=1e-3)
            if Cin % 32 == 0 and Cin >= 64:
                god = go.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().to(DEV)
                wd_lds = K.weight_layout_split(wk, Cout, Cin, 9, 2)                    # [Cin][tap][Cout], LDS-staged kernel
                wd_frag = K.weight_layout_split(wk, Cout, Cin, 9, 2, frag=True)
                g_lds = K.conv_fwd(god, wd_lds, 9, None, B * H * W, Cin, spec.bwd_ix, w_split=True)
                g_frag = K.conv_fwd(god, wd_frag, 9, None, B * H * W, Cin, spec.bwd_ix, w_split=2)
                assert torch.equal(g_frag, g_lds)
                close(g_frag, xr.grad.permute(0, 2, 3, 1).reshape(-1, Cin), rtol=1e-3, atol=1e-4, what="fragment-major dgrad")
            # torch-layout sources (Conv2d [Cout][Cin][kh][kw]) through the multi-job conversion give the same operand bytes
            assert torch.equal(K.weight_layout_split(w.to(DEV).contiguous(), Cout, Cin, 9, 1, frag=True), wf)
        close(stats[:Cout], pre_rows.sum(0), rtol=1e-3, atol=2e-3); close(stats[Cout:], (pre_rows * pre_rows).sum(0), rtol=1e-3, atol=2e-3)
        xd = rows.clone().requires_grad_(True)
        wd = torch.nn.Parameter(w.to(DEV))
        y = A.conv(xd, wd, None, spec, Cout)
        (y * go.permute(0, 2, 3, 1).reshape(-1, Cout).to(DEV)).sum().backward()
        close(xd.grad, xr.grad.permute(0, 2, 3, 1).reshape(-1, Cin), rtol=1e-3, atol=1e-4, what="halo conv dgrad")
    finally:
        K.set_conv_math("f32")


@pytest.mark.parametrize("math", ["f32", "bf16x3"])
def test_shared_weight_and_hooked_weight_gradients_with_deferred_layout(math):
    """The deferred weight-gradient re-layout hands autograd an EMPTY tensor that is filled when the backward pass ends.  A Conv2d
    weight used TWICE in one graph (autograd sums the two gradients before AccumulateGrad) and a weight with a tensor hook (reads
    the gradient mid-pass) must not go through it: both against torch conv2d on the CPU."""
    from radardistill_amd import autograd as A, kernels as K
    g = np.random.default_rng(11)
    B, H, W, C = 2, 12, 16, 64
    x = torch.from_numpy(g.normal(size=(B, C, H, W)).astype(np.float32))
    w = torch.from_numpy((g.normal(size=(C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32))
    w2 = torch.from_numpy((g.normal(size=(C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32))
    wr, w2r = w.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    seen = []
    w2r.register_hook(lambda gr: (seen.append(gr.detach().clone()), gr * 2.0)[1])
    yr = F.conv2d(F.conv2d(F.conv2d(x, wr, None, 1, 1), w2r, None, 1, 1), wr, None, 1, 1)
    yr.square().sum().backward()
    K.set_conv_math(math)
    try:
        spec = A.dense_conv_spec(B, H, W, 3, 3, 1, 1)
        rows = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(DEV)
        wd, w2d = torch.nn.Parameter(w.to(DEV)), torch.nn.Parameter(w2.to(DEV))
        got = []
        w2d.register_hook(lambda gr: (got.append(gr.detach().clone()), gr * 2.0)[1])
        A.begin_step(torch.device(DEV))
        y = A.conv(A.conv(A.conv(rows, wd, None, spec, C), w2d, None, spec, C), wd, None, spec, C)
        y.square().sum().backward()
        torch.cuda.synchronize()
        close(wd.grad, wr.grad, rtol=1e-3, atol=1e-3 * float(wr.grad.abs().max()), what="weight used twice")
        close(got[0], seen[0], rtol=1e-3, atol=1e-3 * float(seen[0].abs().max()), what="gradient seen by the tensor hook")
        close(w2d.grad, w2r.grad, rtol=1e-3, atol=1e-3 * float(w2r.grad.abs().max()), what="hooked weight")
    finally:
        K.set_conv_math("f32")


# ------------------------------------------------------------------------------------------ inference post-processing (8(f) rank 2)
def _clustered_boxes(n, seed):
    g = np.random.default_rng(seed)
    k = max(1, n // 6)
    centres = g.uniform(-40, 40, size=(k, 2))
    c = centres[g.integers(0, k, size=n)] + g.normal(0, 1.2, size=(n, 2))
    boxes = np.concatenate([c, g.normal(0, 0.5, size=(n, 1)), g.uniform(1.5, 5.0, size=(n, 1)), g.uniform(0.8, 2.5, size=(n, 1)),
                            g.uniform(1.0, 2.5, size=(n, 1)), g.uniform(-np.pi, np.pi, size=(n, 1))], axis=1).astype(np.float32)
    scores = g.uniform(0.05, 1.0, size=n).astype(np.float32)
    return torch.from_numpy(boxes), torch.from_numpy(scores)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000, 4200])
def test_rotated_nms_vs_oracle(n):
    """rd_nms_bev (bit matrix + greedy pass both on the device) vs the C restatement: the kept index list is bit-exact."""
    from oracle import post
    from radardistill_amd.pcdet.ops.iou3d_nms import iou3d_nms_utils as U
    boxes, scores = _clustered_boxes(n, 100 + n)
    kept = {}
    for thresh in (0.2, 0.55):
        keep, _ = U.nms_gpu(boxes.to(DEV), scores.to(DEV), thresh)
        ref, _ = post.nms_gpu(boxes, scores, thresh)
        assert keep.dtype == torch.int64 and keep.cpu().tolist() == ref.tolist(), (n, thresh, len(ref))
        kept[thresh] = len(ref)
    if n >= 64:
        keep, _ = U.nms_gpu(boxes.to(DEV), scores.to(DEV), 0.2, pre_maxsize=50)
        assert keep.cpu().tolist() == post.nms_gpu(boxes, scores, 0.2, pre_maxsize=50)[0].tolist()
    if n >= 1000:
        assert 0 < kept[0.2] < kept[0.55] < n        # the test data really exercises suppression


def test_pairwise_iou3d_vs_oracle():
    from oracle import post
    from radardistill_amd.pcdet.ops.iou3d_nms import iou3d_nms_utils as U
    a, _ = _clustered_boxes(70, 7)
    b, _ = _clustered_boxes(33, 7)
    close(U.boxes_iou3d_gpu(a.to(DEV), b.to(DEV)), post.boxes_iou3d(a, b), rtol=1e-4, atol=1e-5)
    assert U.boxes_iou3d_gpu(a[:0].to(DEV), b.to(DEV)).shape == (0, 33)


def test_center_head_eval_decode_golden(golden_dir):
    """Eval path of the head (batched branches -> top-K decode -> rotated NMS on the device) vs the reference head's own output."""
    g = np.load(f"{golden_dir}/g6_decode.npz")
    m = _head()
    m.eval()
    r = np.random.default_rng(26)
    feat = torch.from_numpy(r.normal(0, 1, size=(2, 256, 16, 16)).astype(np.float32))
    with torch.no_grad():
        d = m({"radar_spatial_features_2d": _cl(feat), "batch_size": 2})
    for h, pd in enumerate(m.forward_ret_dict["pred_dicts"]):
        for k, v in pd.items():
            close(v, g[f"pred_{h}_{k}"], what=f"pred {h} {k}")
    def canon(boxes, scores, labels):
        # detections come out in descending-score order per head; scores 1e-6 apart may swap places between two fp32
        # implementations (torch.topk leaves ties open in the reference too), so compare them as a SET: order by (label, x, y)
        key = np.lexsort((np.round(boxes[:, 1], 3), np.round(boxes[:, 0], 3), labels))
        return boxes[key], scores[key], labels[key]

    for b, fd in enumerate(d["final_box_dicts"]):
        assert fd["pred_boxes"].shape == g[f"boxes_{b}"].shape, (fd["pred_boxes"].shape, g[f"boxes_{b}"].shape)
        hb, hs, hl = canon(fd["pred_boxes"].cpu().numpy(), fd["pred_scores"].cpu().numpy(), fd["pred_labels"].cpu().numpy())
        gb, gsc, gl = canon(g[f"boxes_{b}"], g[f"scores_{b}"], g[f"labels_{b}"])
        assert np.array_equal(hl, gl)
        close(hb, gb, what="boxes")
        close(hs, gsc, what="scores")
        # and the order itself is descending in score within each head's block, up to that fp noise
        sc = fd["pred_scores"].cpu().numpy(); lab = fd["pred_labels"].cpu().numpy()
        assert np.all(np.diff(sc)[np.diff(np.searchsorted([1, 2, 4, 6, 7, 9, 11], lab, side="right")) == 0] <= 1e-5)


def test_config0_radar_only_graph_on_the_hip_path_vs_oracle():
    """BASELINE configs[0]: the radar-only graph radar_distill_val.yaml builds (tools/cfgs/radar_distill/radar_distill_val.yaml:67 --
    Radar_DynamicPillarVFESimple2D_Test -> Radar_PillarRes18BackBone8x -> Radar_Distill -> Radar_CenterHead, chained by
    detectors/pillarnet.py:28-46), 1 000 radar points, 128 x 128 BEV, B = 1, eval mode: every map of the six task heads' pred dicts
    from the HIP chain against oracle.pillarnet.forward_radar_only at 1e-3, in both arithmetic modes; the decoded boxes come out of
    the same call.  The module list is derived from the bench yaml exactly as the reference's val yaml differs from its train yaml:
    the teacher keys and DISTILL / FREEZE_PIPELINE removed, the test-time radar VFE that reads `points`."""
    import os
    from radardistill_amd import kernels as K
    from radardistill_amd.data import SyntheticDistillDataset
    from radardistill_amd.pcdet.config import AttrDict, cfg_from_yaml_file
    from radardistill_amd.pcdet.models import build_network, load_data_to_gpu
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c = cfg_from_yaml_file(os.path.join(root, "tools/cfgs/radar_distill/bench_512.yaml"), AttrDict())
    for k in ("VFE", "BACKBONE_3D", "BACKBONE_2D", "DENSE_HEAD", "FREEZE_PIPELINE", "DISTILL"):
        c.MODEL.pop(k, None)
    c.MODEL.RADAR_VFE.NAME = "Radar_DynamicPillarVFESimple2D_Test"
    grid = 128
    pc_range, voxel, gs = bench_geometry(grid)
    c.DATA_CONFIG.POINT_CLOUD_RANGE = pc_range
    c.MODEL.RADAR_BACKBONE_2D.POINT_CLOUD_RANGE = pc_range
    torch.manual_seed(0)
    model = build_network(model_cfg=c.MODEL, num_class=len(c.CLASS_NAMES), dataset=SyntheticDistillDataset.from_cfg(c))
    assert [type(m).__name__ for m in model.module_list] == ["Radar_DynamicPillarVFESimple2D_Test", "Radar_PillarRes18BackBone8x", "Radar_Distill",
                                                              "Radar_CenterHead"]
    sd = model.state_dict(); seeded_fill_(sd, seed=31); model.load_state_dict(sd)
    b = make_batch(batch_size=1, n_lidar=16, n_radar=1000, n_boxes=4, grid=grid, seed=0)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        want = opn.forward_radar_only(state, torch.from_numpy(b["radar_points"]), 1, pc_range, voxel, gs)
    assert len(want) == 6
    model = model.to(DEV).eval()
    for math in ("f32", "bf16x3"):
        K.set_conv_math(math)
        try:
            bd = {"points": b["radar_points"].copy(), "gt_boxes": b["gt_boxes"].copy(), "batch_size": 1}      # the radar-only test set hands its sweep over as `points`
            load_data_to_gpu(bd)
            with torch.no_grad():
                final, recall = model(bd)
            got = model.radar_dense_head.forward_ret_dict["pred_dicts"]
            assert len(got) == 6 and len(final) == 1 and final[0]["pred_boxes"].shape[1] == 9
            for h, (gd, wd) in enumerate(zip(got, want)):
                assert set(gd) == set(wd) == {"center", "center_z", "dim", "rot", "vel", "iou", "hm"}
                for name in wd:
                    assert tuple(gd[name].shape) == tuple(wd[name].shape) == (1, wd[name].shape[1], grid // 8, grid // 8)
                    close(gd[name], wd[name], rtol=1e-3, atol=1e-3 * float(wd[name].abs().max()), what=f"{math} head {h} {name}")
        finally:
            K.set_conv_math("f32")


def test_pillarnet_eval_forward_and_recall():
    """model.eval(): the whole student + teacher chain, decode, NMS and the recall records; the records are re-derived with the
    oracle's pairwise IoU from the predicted boxes."""
    from oracle import post
    grid, B = 128, 2
    model, cfg, pc_range, voxel, gs = _build_pillarnet(grid)
    sd = model.state_dict(); seeded_fill_(sd, seed=77); model.load_state_dict(sd)
    model = model.to(DEV).eval()
    batch = make_batch(batch_size=B, n_lidar=300, n_radar=700, n_boxes=10, grid=grid, seed=5)
    from radardistill_amd.pcdet.models import load_data_to_gpu
    bd = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()}
    load_data_to_gpu(bd)
    with torch.no_grad():
        pred_dicts, recall = model(bd)
    assert len(pred_dicts) == B
    tl = list(cfg.MODEL.POST_PROCESSING.RECALL_THRESH_LIST)
    want = {"gt": 0, **{f"rcnn_{t}": 0 for t in tl}}
    for b in range(B):
        pb = pred_dicts[b]
        assert pb["pred_boxes"].shape[0] == pb["pred_scores"].shape[0] == pb["pred_labels"].shape[0]
        assert pb["pred_boxes"].shape[0] <= 6 * cfg.MODEL.RADAR_DENSE_HEAD.POST_PROCESSING.NMS_CONFIG.NMS_POST_MAXSIZE
        if pb["pred_labels"].numel():
            assert int(pb["pred_labels"].min()) >= 1 and int(pb["pred_labels"].max()) <= 10
        rec = post.recall_record(pb["pred_boxes"].cpu(), torch.from_numpy(batch["gt_boxes"][b]), tl)
        for k in want:
            want[k] += rec[k]
    assert {k: recall[k] for k in want} == want


def test_split_format_round_trip_and_presplit_conv():
    """rd_split_bf16: hi + lo reproduces x to ~2^-17; a conv on pre-split operands equals the in-loop split bit for bit."""
    from radardistill_amd import autograd as A, kernels as K
    g = np.random.default_rng(3)
    x = torch.from_numpy(g.normal(size=(1000, 64)).astype(np.float32) * 7).to(DEV)
    xs = K.split_bf16(x)
    raw = xs.view(torch.bfloat16).view(-1, 8).float()                 # [hi0..hi3 | lo0..lo3] per group of 4
    rec = (raw[:, :4] + raw[:, 4:]).reshape(x.shape)
    assert float(((rec - x).abs() / x.abs().clamp_min(1e-20)).max()) < 2 ** -15
    assert torch.equal(raw[:, :4].reshape(x.shape), x.to(torch.bfloat16).float())
    K.set_conv_math("bf16x3")
    try:
        spec = A.dense_conv_spec(2, 16, 16, 3, 3, 1, 1)
        xi = torch.from_numpy(g.normal(size=(512, 64)).astype(np.float32)).to(DEV)
        w = torch.from_numpy(g.normal(size=(96, 9, 64)).astype(np.float32) * 0.1).to(DEV)
        ref = K.conv_fwd(xi, w, 9, None, 512, 96, spec.fwd_ix)
        ws = K.weight_layout_split(w, 96, 64, 9, 0)
        for a_s, b_s in ((True, True), (True, False), (False, True)):
            out = K.conv_fwd(K.split_bf16(xi) if a_s else xi, ws if b_s else w, 9, None, 512, 96, spec.fwd_ix, in_split=a_s, w_split=b_s)
            if a_s:      # a split activation operand runs on the gathered kernel, the fp32 one on the halo kernel: other summation order
                close(out, ref, rtol=1e-5, atol=1e-6, what=str((a_s, b_s)))
            else:
                assert torch.equal(out, ref), (a_s, b_s)
        s2 = A.dense_conv_spec(2, 16, 16, 3, 3, 2, 1)                   # stride 2: gathered kernel for every operand format -> bit-equal
        ref2 = K.conv_fwd(xi, w, 9, None, 128, 96, s2.fwd_ix)
        for a_s, b_s in ((True, True), (True, False), (False, True)):
            out = K.conv_fwd(K.split_bf16(xi) if a_s else xi, ws if b_s else w, 9, None, 128, 96, s2.fwd_ix, in_split=a_s, w_split=b_s)
            assert torch.equal(out, ref2), (a_s, b_s)
        go = torch.from_numpy(g.normal(size=(512, 96)).astype(np.float32)).to(DEV)
        A.begin_step(torch.device(DEV))
        gw_ref = K.conv_wgrad(xi, go, 9, spec.fwd_ix).clone()
        gw = K.conv_wgrad(K.split_bf16(xi), K.split_bf16(go), 9, spec.fwd_ix, in_split=True, go_split=True)
        close(gw, gw_ref, rtol=1e-5, atol=1e-6)                        # same products, atomics in another order
    finally:
        K.set_conv_math("f32")


@pytest.mark.parametrize("B,hw,C", [(2, 256, 1024), (3, 37, 64), (1, 1000, 260)])
def test_fused_gelu_grn_forward_backward(B, hw, C):
    """rd_gelu_grn_{fwd,bwd} vs nn.GELU + GRN (Basicblock_convn.py:46-60) under torch autograd on the CPU."""
    from radardistill_amd import autograd as A
    from radardistill_amd.pcdet.ops.basicblock.Basicblock_convn import GRN
    g = np.random.default_rng(B * 7 + C)
    z = torch.from_numpy(g.normal(size=(B, hw, 1, C)).astype(np.float32))
    go = torch.from_numpy(g.normal(size=(B, hw, 1, C)).astype(np.float32))
    grn = GRN(C)
    with torch.no_grad():
        grn.gamma.copy_(torch.from_numpy(g.normal(size=(1, 1, 1, C)).astype(np.float32)))
        grn.beta.copy_(torch.from_numpy(g.normal(size=(1, 1, 1, C)).astype(np.float32)))
    zr = z.clone().requires_grad_(True)
    ref = grn(torch.nn.functional.gelu(zr))
    (ref * go).sum().backward()
    import copy
    gd = copy.deepcopy(grn).to(DEV)
    gd.gamma.grad = gd.beta.grad = None
    zd = z.reshape(B * hw, C).to(DEV).requires_grad_(True)
    out = A.gelu_grn(zd, gd, B)
    close(out, ref.reshape(B * hw, C), rtol=1e-4, atol=1e-5, what="gelu+grn forward")
    (out * go.reshape(B * hw, C).to(DEV)).sum().backward()
    close(zd.grad, zr.grad.reshape(B * hw, C), rtol=1e-3, atol=1e-4, what="grad z")
    close(gd.gamma.grad, grn.gamma.grad, rtol=1e-3, atol=1e-4, what="grad gamma")
    close(gd.beta.grad, grn.beta.grad, rtol=1e-3, atol=1e-4, what="grad beta")


# ------------------------------------------------------------------------------------------ BASELINE full sizes: size-independent properties
def test_full_size_training_step_two_arithmetic_modes_agree():
    """BASELINE configs[3] at its real size (B = 8, 512 x 512 BEV, 35 k + 2 k points, 30 boxes; too big for the CPU oracle):
    the exact-fp32 kernels and the bf16x3 kernels are two independent implementations of every convolution / weight gradient --
    one training step in each mode from the same state must agree on the loss, every tb entry and the clipped gradient norm."""
    from radardistill_amd import kernels as K
    from radardistill_amd.pcdet.models import model_fn_decorator
    from radardistill_amd.train import build_optimizer
    model, cfg, pc_range, voxel, gs = _build_pillarnet(512)
    sd = model.state_dict(); seeded_fill_(sd, seed=5); model.load_state_dict(sd)
    model = model.to(DEV).train()
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    batch = make_batch(batch_size=8, n_lidar=35000, n_radar=2000, n_boxes=30, grid=512, seed=2)
    out = {}
    try:
        for mode in ("f32", "bf16x3"):
            K.set_conv_math(mode)
            model.load_state_dict(init)
            opt = build_optimizer(model, cfg.OPTIMIZATION)
            opt.zero_grad()
            loss, tb, _ = model_fn_decorator()(model, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()})
            loss.backward()
            norm = opt.step()
            g = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
            assert torch.isfinite(g).all() and torch.isfinite(loss)
            out[mode] = (float(loss.detach()), {k: float(v) for k, v in tb.items()}, float(norm[0]), g)
    finally:
        K.set_conv_math("f32")
    (la, ta, na, ga), (lb, tb_, nb, gb) = out["f32"], out["bf16x3"]
    assert abs(la - lb) <= 1e-3 * abs(la), (la, lb)
    for k in ta:
        assert abs(ta[k] - tb_[k]) <= 2e-3 * abs(ta[k]) + 1e-5, (k, ta[k], tb_[k])
    rel = float((ga - gb).norm() / ga.norm())
    print("full-size step: loss", la, lb, "grad norm", na, nb, "gradient rel L2 between modes", rel)
    assert abs(na - nb) <= 5e-2 * na and rel <= 1.5e-1          # ReLU sign flips between two roundings, see test_full_distillation_step_vs_oracle


def test_dense_enc_at_1024_bev_size_modes_agree_and_linear():
    """BASELINE configs[4] shapes (1024 x 1024 BEV -> x_conv4 (B,256,128,128), x_conv5 (B,256,64,64)), eval mode (a fixed affine map
    up to ReLU): fp32 vs bf16x3 kernels agree, and the map commutes with a batch permutation (no cross-sample leakage)."""
    from radardistill_amd import kernels as K
    from radardistill_amd.pcdet.config import AttrDict
    from radardistill_amd.pcdet.models.backbones_2d import __all__ as REG
    m = REG["BaseBEVBackboneV2"](AttrDict(BEV_CFG), input_channels=256)
    sd = m.state_dict(); seeded_fill_(sd, seed=12); m.load_state_dict(sd); m = m.to(DEV).eval()
    g = np.random.default_rng(4)
    x4 = torch.from_numpy((g.normal(size=(2, 256, 128, 128)) * (g.uniform(size=(2, 1, 128, 128)) < 0.3)).astype(np.float32))
    x5 = torch.from_numpy(g.normal(size=(2, 256, 64, 64)).astype(np.float32))
    res = {}
    try:
        for mode in ("f32", "bf16x3"):
            K.set_conv_math(mode)
            with torch.no_grad():
                d = m({"multi_scale_2d_features": {"x_conv4": _cl(x4), "x_conv5": _cl(x5)}})
                dp = m({"multi_scale_2d_features": {"x_conv4": _cl(x4.flip(0)), "x_conv5": _cl(x5.flip(0))}})
            res[mode] = (d["spatial_features_2d"].clone(), d["spatial_features_2d_8x"].clone())
            assert torch.equal(dp["spatial_features_2d"].flip(0), d["spatial_features_2d"])
    finally:
        K.set_conv_math("f32")
    close(res["bf16x3"][0], res["f32"][0], rtol=1e-3, atol=1e-4); close(res["bf16x3"][1], res["f32"][1], rtol=1e-3, atol=1e-4)


# ------------------------------------------------------------------------------------------ the reference yaml's own geometry
def _build_real_geometry():
    """radar_distill_train.yaml:6,63: +-54 m at 0.075 m pillars -> 1440 x 1440 grid, 180 x 180 head map (the yaml itself does not
    travel to the GPU box: same MODEL block from tools/cfgs/radar_distill/bench_512.yaml, geometry overridden)."""
    import os
    from radardistill_amd.data import SyntheticDistillDataset
    from radardistill_amd.pcdet.config import AttrDict, cfg_from_yaml_file
    from radardistill_amd.pcdet.models import build_network
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c = cfg_from_yaml_file(os.path.join(root, "tools/cfgs/radar_distill/bench_512.yaml"), AttrDict())
    pc_range = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]
    c.DATA_CONFIG.POINT_CLOUD_RANGE = pc_range
    c.DATA_CONFIG.VOXEL_SIZE = [0.075, 0.075, 0.2]
    c.MODEL.RADAR_BACKBONE_2D.POINT_CLOUD_RANGE = pc_range
    c.MODEL.RADAR_BACKBONE_2D.VOXEL_SIZE = [0.075, 0.075, 8.0]
    c.MODEL.RADAR_BACKBONE_2D.GRID_SIZE = [1440, 1440, 1]
    ds = SyntheticDistillDataset.from_cfg(c)
    assert list(ds.grid_size) == [1440, 1440, 40]
    torch.manual_seed(0)
    m = build_network(model_cfg=c.MODEL, num_class=len(c.CLASS_NAMES), dataset=ds)
    sd = m.state_dict(); seeded_fill_(sd, seed=78); m.load_state_dict(sd)
    return m.to(DEV), pc_range, [0.075, 0.075, 0.2], ds.grid_size


def test_training_step_at_the_reference_yaml_geometry():
    """One distillation step at 1440 x 1440 / 0.075 m (B = 2), checked through size-independent properties: pillar coordinates
    bit-exact vs the oracle voxeliser (division by 0.075, int(-54.0) origin), the two arithmetic modes agree on every loss term,
    swapping the two samples leaves the (batch-mean) loss unchanged, every trainable parameter receives a finite gradient."""
    from oracle import vfe as ovfe
    from radardistill_amd import kernels as K
    from radardistill_amd.pcdet.models import model_fn_decorator
    model, pc_range, voxel, gs = _build_real_geometry()
    B = 2
    batch = make_batch(batch_size=B, n_lidar=35000, n_radar=2000, n_boxes=30, grid=540, seed=21)      # +-54 m
    fn = model_fn_decorator()

    def run(bd):
        model.zero_grad(set_to_none=True)
        model.train()
        loss, tb, _ = fn(model, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in bd.items()})
        loss.backward()
        return float(loss), {k: float(v) for k, v in tb.items()}, {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.requires_grad}

    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    loss_a, tb_a, grads = run(batch)
    assert np.isfinite(loss_a)
    for k, g in grads.items():
        assert g is not None and bool(torch.isfinite(g).all()), k
    # (1) pillar coordinates of both branches at this geometry, bit-exact (eval pass keeps them in the dict)
    model.eval()
    bd = {k: (torch.from_numpy(v).to(DEV) if isinstance(v, np.ndarray) else v) for k, v in batch.items()}
    with torch.no_grad():
        for mod, key, pre in ((model.vfe, "points", ""), (model.radar_vfe, "radar_points", "radar_")):
            out = mod({key: bd[key], "batch_size": B})
            _, _, unq, _, _ = ovfe.voxelize(torch.from_numpy(batch[key]), pc_range, voxel, (1440, 1440))
            unq = unq.long()                   # key = b * 1440^2 + cx * 1440 + cy, sorted: the reference's pillar order (:243-248)
            ref = torch.stack((unq // (1440 * 1440), unq % 1440, (unq % (1440 * 1440)) // 1440), dim=1).int()       # (b, y, x)
            got = out[pre + "pillar_coords"].cpu()
            assert got.shape == ref.shape and torch.equal(got, ref), key
    # (2) bf16x3 arithmetic on the same weights and batch
    model.load_state_dict(state0)
    K.set_conv_math("bf16x3")
    try:
        loss_b, tb_b, _ = run(batch)
    finally:
        K.set_conv_math("f32")
    assert abs(loss_a - loss_b) <= 1e-3 * abs(loss_a), (loss_a, loss_b)
    for k in tb_a:
        assert abs(tb_a[k] - tb_b[k]) <= 2e-3 * abs(tb_a[k]) + 1e-5, (k, tb_a[k], tb_b[k])
    # (3) sample permutation
    model.load_state_dict(state0)
    perm = {"batch_size": B, "gt_boxes": batch["gt_boxes"][::-1].copy()}
    for key in ("points", "radar_points"):
        p = batch[key]
        parts = [p[p[:, 0] == b].copy() for b in range(B)][::-1]
        for b, part in enumerate(parts):
            part[:, 0] = b
        perm[key] = np.concatenate(parts, 0)
    loss_c, tb_c, _ = run(perm)
    assert abs(loss_a - loss_c) <= 2e-4 * abs(loss_a), (loss_a, loss_c)


def test_afd_nan_edge_gradient_contract():
    """The documented deviation in the AFD edge case.  When no (radar-active, LiDAR-inactive) cell exists the reference's
    `mask_ir * (N_ar / N_ir)` is 0 * inf: the forward value is NaN (reproduced, fixture g3) and torch autograd then writes NaN into
    the WHOLE radar-map gradient, which ends the reference run at the next optimizer step.  rd_afd_bwd instead returns the gradient
    of the terms that are defined -- the active-region MSE and the mask L1 -- with nothing from the empty region: finite everywhere,
    equal to autograd on those terms.  (Reachable: ~360 steps of overfitting two synthetic batches, tools/diag/nan_hunt.py.)"""
    m = _radar_distill()
    r = np.random.default_rng(41)
    lid = torch.from_numpy(np.abs(r.normal(0.2, 1, size=(2, 256, 16, 16))).astype(np.float32) + 1.0)      # every cell LiDAR-active
    rad = torch.from_numpy(r.normal(0.0, 1, size=(2, 256, 16, 16)).astype(np.float32))
    rd = _cl(rad).requires_grad_(True)
    f, ml = m.low_loss(_cl(lid), rd)
    assert torch.isnan(f).item() and torch.isfinite(ml).item()
    (f * 2.5 + ml * 2.5).backward()                       # the weights get_loss applies: 5 * 0.5
    assert bool(torch.isfinite(rd.grad).all())
    # the reference expression, term by term: its gradient is NaN everywhere ...
    ro = rad.clone().requires_grad_(True)
    fo, mo = obev.low_loss(lid, ro)
    assert torch.isnan(fo).item()
    (fo * 2.5 + mo * 2.5).backward()
    assert bool(torch.isnan(ro.grad).all())
    # ... and the defined terms alone give what the kernel returns
    rr = rad.clone().requires_grad_(True)
    lidar_mask = (lid.sum(1, keepdim=True) > 0).float()
    radar_sum = rr.sum(1, keepdim=True)
    m_ar = (((radar_sum > 0).float() + 0.5 * lidar_mask) == 1.5).float()
    defined = 3e-4 * (F.mse_loss(rr, lid, reduction="none") * m_ar).sum() / 2 * 2.5 + F.l1_loss(torch.sigmoid(radar_sum), lidar_mask) * 2.5
    defined.backward()
    close(rd.grad, rr.grad, rtol=1e-3, atol=1e-6 * float(rr.grad.abs().max()), what="AFD gradient in the NaN edge")


def test_dense_graph_replay_is_bit_identical_to_the_eager_step():
    """MODEL.DENSE_GRAPH: the static dense section (conv5, DenseEnc, CMA, heads, targets, losses + backward) replayed as ONE HIP graph
    against the eager path: three optimizer steps over two alternating batches (different point counts, different numbers of boxes)
    with the library's reductions in fixed order -> every loss term, every gradient and every updated parameter bit-identical."""
    import os
    from radardistill_amd import kernels as K
    from radardistill_amd.pcdet.models import model_fn_decorator
    from radardistill_amd.train import build_optimizer, build_scheduler
    grid, B = 128, 2
    # (sparse LiDAR sweeps: with every 8x cell LiDAR-active the AFD term is NaN by definition, which compares unequal to itself)
    batches = [make_batch(batch_size=B, n_lidar=300, n_radar=700, n_boxes=nb, grid=grid, seed=s_) for s_, nb in ((5, 10), (6, 17))]
    fn = model_fn_decorator()

    def run(graph):
        model, cfg, pc_range, voxel, gs = _build_pillarnet(grid)
        sd = model.state_dict(); seeded_fill_(sd, seed=77); model.load_state_dict(sd)
        model = model.to(DEV)
        model.model_cfg['DENSE_GRAPH'] = graph
        opt = build_optimizer(model, cfg.OPTIMIZATION)
        sched, _ = build_scheduler(opt, 10, 1, -1, cfg.OPTIMIZATION)
        out = []
        model.train()
        for it in range(3):
            sched.step(it)
            opt.zero_grad()
            loss, tb, _ = fn(model, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batches[it % 2].items()})
            loss.backward()
            grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
            opt.step()
            torch.cuda.synchronize()
            out.append((loss.detach().clone(), {k: torch.as_tensor(v).detach().clone() for k, v in tb.items()}, grads,
                        {k: v.detach().clone() for k, v in model.state_dict().items()}))
        return out

    from radardistill_amd import autograd as A
    prev = os.environ.pop("RD_DENSE_GRAPH", None)
    # Bit identity needs the SAME autograd graph on both sides: where a tensor has several consumers the engine adds their gradients in
    # arrival order.  The rows shortcut (autograd.ROWS_SHORTCUT: consecutive layers pass the rows tensor along, no view / permute nodes
    # between them) does not apply inside a capture -- its tensors are static and outlive the step -- so the eager side runs without it
    # here; with it the two sides differ in the last bit of a few gradients (summation order at the fan-outs of x_conv4 / de_8x).
    shortcut = A.ROWS_SHORTCUT[0]
    A.ROWS_SHORTCUT[0] = False
    K.set_deterministic(True)
    try:
        eager, graphed = run(False), run(True)
    finally:
        K.set_deterministic(False)
        A.ROWS_SHORTCUT[0] = shortcut
        if prev is not None:
            os.environ["RD_DENSE_GRAPH"] = prev
    for it, ((l0, t0, g0, s0), (l1, t1, g1, s1)) in enumerate(zip(eager, graphed)):
        assert torch.equal(l0, l1) and bool(torch.isfinite(l0)), (it, float(l0), float(l1))
        assert t0.keys() == t1.keys()
        for k in t0:
            a, b = t0[k].float().cpu().reshape(-1), t1[k].float().cpu().reshape(-1)
            assert torch.equal(a, b) or (bool(torch.isnan(a).all()) and bool(torch.isnan(b).all())), (it, k, a, b)
        assert g0.keys() == g1.keys(), (it, set(g0) ^ set(g1))
        for k in g0:
            assert torch.equal(g0[k], g1[k]), (it, "grad", k)
        for k in s0:
            assert torch.equal(s0[k], s1[k]), (it, "state", k)
