"""GPU parity tests (run with -m gpu on an MI355X): every HIP kernel through the C ABI vs the CPU oracle / torch-CPU.
Tolerance: bit-exact for indices / rulebooks; rtol 1e-3 (north_star) for fp32 features and gradients."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import sparse as osp, vfe as ovfe            # noqa: E402  (checker only)
from radardistill_amd.synthetic import bench_geometry, make_batch   # noqa: E402
from tests.seeded import seeded_fill_                    # noqa: E402

DEV = "cuda:0"


def _mods():
    from radardistill_amd import autograd as A, kernels as K, sparse as SP
    return A, K, SP


def close(a, b, rtol=1e-3, atol=1e-4, what=""):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    scale = max(1.0, float(np.abs(b).max())) if b.size else 1.0
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * scale, err_msg=what)


def rand_sites(rng, B, H, W, n):
    keys = np.sort(rng.choice(B * H * W, size=n, replace=False))
    return np.stack([keys // (H * W), (keys // W) % H, keys % W], axis=1).astype(np.int32)


@pytest.mark.parametrize("act,res", [(1, False), (1, True), (2, False)])
def test_fused_conv_bn_act_node_equals_separate_nodes(act, res):
    """autograd._ConvBNActFn (conv -> train-mode BatchNorm -> (+residual) -> act as ONE autograd node) against the two separate nodes
    it replaces (A.conv with fused statistics + A.bn_act_train): same kernels, so outputs, running statistics and every gradient
    (input, weight, bias, gamma, beta, residual) agree up to the order of the atomics."""
    A, K, SP = _mods()
    rng = np.random.default_rng(5 + act + 2 * res)
    B, H, W, Cin, Cout = 2, 12, 10, 64, 96
    spec = A.dense_conv_spec(B, H, W, 3, 3, 1, 1)
    x0 = torch.from_numpy(rng.normal(size=(B * H * W, Cin)).astype(np.float32)).to(DEV)
    r0 = torch.from_numpy(rng.normal(size=(B * H * W, Cout)).astype(np.float32)).to(DEV) if res else None
    go = torch.from_numpy(rng.normal(size=(B * H * W, Cout)).astype(np.float32)).to(DEV)
    conv = torch.nn.Conv2d(Cin, Cout, 3, 1, 1, bias=True).to(DEV)
    bn = torch.nn.BatchNorm2d(Cout, eps=1e-3, momentum=0.01).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, Cout).astype(np.float32)))
        bn.bias.copy_(torch.from_numpy(rng.normal(size=Cout).astype(np.float32)))
    out = {}
    for fused in (False, True):
        for p_ in list(conv.parameters()) + list(bn.parameters()):
            p_.grad = None
        bn.running_mean.zero_(); bn.running_var.fill_(1.0)
        x = x0.clone().requires_grad_(True)
        r = r0.clone().requires_grad_(True) if res else None
        A.begin_step(torch.device(DEV))
        if fused:
            y = A.conv_bn_act_train(x, conv.weight, conv.bias, spec, Cout, bn, r, act)
        else:
            st = A.zeros_stats(2 * Cout, x.device)
            y = A.bn_act_train(A.conv(x, conv.weight, conv.bias, spec, Cout, st), bn, r, act=act, stats=st)
        (y * go).sum().backward()
        torch.cuda.synchronize()
        out[fused] = [y.detach(), x.grad, conv.weight.grad.clone(), conv.bias.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(),
                      bn.running_mean.clone(), bn.running_var.clone()] + ([r.grad] if res else [])
    for i, (a, b) in enumerate(zip(out[True], out[False])):
        if i == 3:
            # gradient of the conv bias in front of the BatchNorm: identically 0 (the BatchNorm subtracts the batch mean).  The fused
            # node returns the exact value; the separate conv node cannot know about the BatchNorm, sums the rows numerically and gets
            # rounding noise, ~1e-7 of the weight gradient's scale
            assert float(a.abs().max()) == 0.0
            assert float(b.abs().max()) <= 1e-5 * float(out[False][2].abs().max())
            continue
        close(a, b, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("math", ["f32", "bf16x3"])
def test_composite_layer_calls_equal_the_per_launch_path(math):
    """rd_conv_bn_act_fwd / rd_conv_bn_act_bwd (one library call per layer and direction, autograd.COMPOSITE) issue exactly the
    launches of the per-launch path: with the reductions in fixed order every output, running statistic and gradient is bit-identical,
    for a sparse layer with a residual and a dense 3x3 layer, weight gradient on the side stream."""
    A, K, SP = _mods()
    rng = np.random.default_rng(5)
    B, H, W, C = 2, 24, 32, 64
    x = torch.from_numpy(rng.normal(size=(B * H * W, C)).astype(np.float32)).to(DEV)
    res = torch.from_numpy(rng.normal(size=(B * H * W, C)).astype(np.float32)).to(DEV)
    w = torch.from_numpy((rng.normal(size=(C, C, 3, 3)) / 24).astype(np.float32))
    go = torch.from_numpy(rng.normal(size=(B * H * W, C)).astype(np.float32)).to(DEV)
    K.set_conv_math(math)
    K.set_deterministic(True)
    prev = A.COMPOSITE[0]
    try:
        out = {}
        for comp in (False, True):
            A.COMPOSITE[0] = comp
            conv = torch.nn.Conv2d(C, C, 3, padding=1, bias=True).to(DEV)
            bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).to(DEV).train()
            with torch.no_grad():
                conv.weight.copy_(w); conv.bias.fill_(0.25); bn.weight.fill_(1.5); bn.bias.fill_(-0.1)
            xd, rd = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
            A.begin_step(torch.device(DEV))
            spec = A.dense_conv_spec(B, H, W, 3, 3, 1, 1)
            y1 = A.conv_bn_act_train(xd, conv.weight, conv.bias, spec, C, bn, residual=rd, act=1)
            y2 = A.conv_bn_act_train(y1, conv.weight, conv.bias, spec, C, bn, residual=None, act=1)          # the weight is used twice
            (y2 * go).sum().backward()
            torch.cuda.synchronize()
            out[comp] = [t.detach().clone() for t in (y1, y2, xd.grad, rd.grad, conv.weight.grad, conv.bias.grad, bn.weight.grad, bn.bias.grad,
                                                      bn.running_mean, bn.running_var)]
        for a, b in zip(out[False], out[True]):
            assert torch.equal(a, b)
    finally:
        A.COMPOSITE[0] = prev
        K.set_deterministic(False)
        K.set_conv_math("f32")


@pytest.mark.parametrize("xmajor", [False, True])
def test_rankgrid_downsample_from_grid_equals_from_coords(xmajor):
    """rd_rankgrid_downsample_grid (marks the SparseConv2d(k3, s2, p1) output set from the input rank GRID, no host-side row count)
    produces the same rank grid, word for word, as rd_rankgrid_downsample on the coordinate list; chained three levels deep."""
    A, K, SP = _mods()
    rng = np.random.default_rng(11)
    B, H, W = 3, 37, 50
    sites = rand_sites(rng, B, H, W, 900)
    if xmajor:
        sites = sites[np.lexsort((sites[:, 1], sites[:, 2], sites[:, 0]))]          # (b, x, y) key order
    coords = torch.from_numpy(sites).to(DEV)
    rg = K.rankgrid_from_coords(coords, B, H, W, xmajor)
    h, w, xm = H, W, xmajor
    for _ in range(3):
        ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        ref = K.rankgrid_downsample(coords, B, ho, wo)
        got = K.rankgrid_downsample_grid(rg, B, h, w, xm, ho, wo)
        assert torch.equal(ref, got)
        n = int(K.rankgrid_count_tensor(got, B * ho * wo))
        assert n > 0
        coords = K.rankgrid_coords(got, B, ho, wo, False, n)
        rg, h, w, xm = got, ho, wo, False


def test_library_reports_gfx950():
    from radardistill_amd import native
    assert native.lib().rd_device_ok() == 1


@pytest.mark.parametrize("H,W,n", [(64, 48, 500), (33, 31, 200), (16, 16, 0), (8, 8, 128)])
def test_rulebooks_bit_exact(H, W, n):
    A, K, SP = _mods()
    rng = np.random.default_rng(H * 1000 + n)
    B = 3
    idx = rand_sites(rng, B, H, W, n)
    coords = torch.from_numpy(idx).to(DEV)
    t = SP.SparseConvTensor(torch.zeros((n, 32), device=DEV), coords, [H, W], B)
    nbr = t._level.subm_spec().fwd_nbr if n else None
    ref = osp.subm_rulebook(idx, (H, W))
    if n:
        assert np.array_equal(nbr.cpu().numpy(), ref)
    lvl, spec = t._level.down()
    oidx, oshape, snbr = osp.strided_rulebook(idx, (H, W))
    assert (lvl.H, lvl.W) == tuple(oshape)
    assert np.array_equal(lvl.coords.cpu().numpy(), oidx)            # canonical (b,y,x)-sorted output sites
    if oidx.shape[0]:
        assert np.array_equal(spec.fwd_nbr.cpu().numpy(), snbr)
        # transposed table: nbrT[i][t] == o  <=>  nbr[o][t] == i
        nbrT = spec.bwd_nbr.cpu().numpy()
        exp = -np.ones_like(nbrT)
        o, tt = np.nonzero(snbr >= 0)
        exp[snbr[o, tt], tt] = o
        assert np.array_equal(nbrT, exp)


def test_rulebook_xmajor_order_matches_voxelizer():
    """pillar rows come in (b, cx, cy) key order; lookups must honour that linearisation."""
    A, K, SP = _mods()
    rng = np.random.default_rng(5)
    B, H, W = 2, 40, 24
    idx = rand_sites(rng, B, H, W, 300)
    order = np.lexsort((idx[:, 1], idx[:, 2], idx[:, 0]))           # sort by (b, x, y)
    idx_x = idx[order]
    t = SP.SparseConvTensor(torch.zeros((300, 32), device=DEV), torch.from_numpy(idx_x).to(DEV), [H, W], B)
    assert t._level.xmajor
    assert np.array_equal(t._level.subm_spec().fwd_nbr.cpu().numpy(), osp.subm_rulebook(idx_x, (H, W)))
    lvl, spec = t._level.down()
    oidx, _, snbr = osp.strided_rulebook(idx_x, (H, W))
    assert np.array_equal(lvl.coords.cpu().numpy(), oidx) and np.array_equal(spec.fwd_nbr.cpu().numpy(), snbr)


def _vfe_module(tag, nfeat, grid, seed=11):
    from radardistill_amd.pcdet.models.backbones_3d.vfe import __all__ as REG
    from radardistill_amd.pcdet.config import AttrDict
    pc_range, voxel, gs = bench_geometry(grid)
    cfg = AttrDict(WITH_DISTANCE=False, USE_ABSLOTE_XYZ=True, USE_CLUSTER_XYZ=True, USE_NORM=True, NUM_FILTERS=[32])
    name = "Radar_DynamicPillarVFESimple2D" if tag == "radar" else "DynamicPillarVFESimple2D"
    m = REG[name](model_cfg=cfg, num_point_features=nfeat, voxel_size=voxel, grid_size=gs, point_cloud_range=pc_range)
    sd = m.state_dict(); seeded_fill_(sd, seed=seed); m.load_state_dict(sd)
    return m.to(DEV), pc_range, voxel, gs


@pytest.mark.parametrize("tag,key,nfeat", [("radar", "radar_points", 6), ("lidar", "points", 5)])
def test_vfe_golden_from_reference(golden_dir, tag, key, nfeat):
    """HIP VFE vs the fixture produced by the reference's own module (tests/golden/make_golden.py:g1_vfe)."""
    g = np.load(f"{golden_dir}/g1_vfe.npz")
    m, pc_range, voxel, gs = _vfe_module(tag, nfeat, 128)
    batch = make_batch(batch_size=2, n_lidar=2000, n_radar=1000, n_boxes=4, grid=128, seed=1)
    pts = torch.from_numpy(batch[key]).clone()
    pts[:5, 1] = torch.tensor([pc_range[3] + 0.05, pc_range[0] - 0.01, pc_range[3], pc_range[0], 0.0])
    pre = "radar_" if tag == "radar" else ""
    for mode in ("eval", "train"):
        m.train(mode == "train")
        with torch.no_grad():
            bd = m({key: pts.to(DEV), "batch_size": 2})
        assert np.array_equal(bd[pre + "pillar_coords"].cpu().numpy(), g[f"{tag}_{mode}_coords"])   # bit-exact
        close(bd[pre + "pillar_features"], g[f"{tag}_{mode}_features"], what=f"{tag} {mode}")
    close(m.pfn_layers[0].norm.running_mean, g[f"{tag}_running_mean_after"])
    close(m.pfn_layers[0].norm.running_var, g[f"{tag}_running_var_after"])


def test_vfe_backward_vs_oracle():
    m, pc_range, voxel, gs = _vfe_module("radar", 6, 128, seed=3)
    batch = make_batch(batch_size=2, n_lidar=16, n_radar=1500, n_boxes=1, grid=128, seed=7)
    pts = torch.from_numpy(batch["radar_points"])
    m.train()
    bd = m({"radar_points": pts.to(DEV), "batch_size": 2})
    f = bd["radar_pillar_features"]
    gw = torch.from_numpy(np.random.default_rng(0).normal(size=tuple(f.shape)).astype(np.float32))
    (f * gw.to(DEV)).sum().backward()
    st = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    seeded_fill_(st, seed=3)
    for k in ("pfn_layers.0.linear.weight", "pfn_layers.0.norm.weight", "pfn_layers.0.norm.bias"):
        st[k].requires_grad_(True)
    out = ovfe.dynamic_pillar_vfe(pts, st, "", pc_range, voxel, gs, training=True)
    close(f, out["pillar_features"])
    (out["pillar_features"] * gw).sum().backward()
    close(m.pfn_layers[0].linear.weight.grad, st["pfn_layers.0.linear.weight"].grad, atol=2e-4)
    close(m.pfn_layers[0].norm.weight.grad, st["pfn_layers.0.norm.weight"].grad, atol=2e-4)
    close(m.pfn_layers[0].norm.bias.grad, st["pfn_layers.0.norm.bias"].grad, atol=2e-4)


def test_vfe_empty_and_all_outside():
    m, pc_range, voxel, gs = _vfe_module("radar", 6, 128)
    m.eval()
    pts = torch.zeros((10, 7)); pts[:, 1] = 1e4                      # every point outside the range
    bd = m({"radar_points": pts.to(DEV), "batch_size": 1})
    assert bd["radar_pillar_features"].shape == (0, 32) and bd["radar_pillar_coords"].shape == (0, 3)
    bd = m({"radar_points": torch.zeros((0, 7), device=DEV), "batch_size": 1})
    assert bd["radar_pillar_features"].shape == (0, 32)


@pytest.mark.parametrize("Cin,Cout", [(32, 32), (64, 128), (128, 256), (32, 27), (256, 3)])
def test_sparse_conv_forward_and_backward(Cin, Cout):
    A, K, SP = _mods()
    rng = np.random.default_rng(Cin + Cout)
    B, H, W, n = 2, 40, 36, 700
    idx = rand_sites(rng, B, H, W, n)
    feats = torch.from_numpy(rng.normal(size=(n, Cin)).astype(np.float32))
    w = torch.from_numpy((rng.normal(size=(Cout, 3, 3, Cin)) / np.sqrt(9 * Cin)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))
    for subm in (True, False):
        conv = (SP.SubMConv2d if subm else SP.SparseConv2d)(Cin, Cout, 3, stride=1 if subm else 2, padding=1, bias=True).to(DEV)
        with torch.no_grad():
            conv.weight.copy_(w); conv.bias.copy_(b)
        fd = feats.to(DEV).requires_grad_(True)
        t = SP.SparseConvTensor(fd, torch.from_numpy(idx).to(DEV), [H, W], B)
        out = conv(t)
        fr = feats.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
        if subm:
            nbr = osp.subm_rulebook(idx, (H, W))
        else:
            oidx, _, nbr = osp.strided_rulebook(idx, (H, W))
            assert np.array_equal(out.indices.cpu().numpy(), oidx)
        ref = osp.sparse_conv(fr, nbr, wr, br)
        close(out.features, ref, what=f"fwd subm={subm}")
        go = torch.from_numpy(rng.normal(size=tuple(ref.shape)).astype(np.float32))
        (out.features * go.to(DEV)).sum().backward()
        (ref * go).sum().backward()
        close(fd.grad, fr.grad, what="dgrad")
        close(conv.weight.grad, wr.grad, atol=2e-4, what="wgrad")
        close(conv.bias.grad, br.grad, atol=2e-4, what="bias grad")


@pytest.mark.parametrize("Cin,Cout,n", [(64, 128, 700), (128, 128, 2500), (256, 256, 150), (64, 64, 5000), (128, 96, 1)])
def test_sparse_conv_fragment_major_kernel_bf16x3(Cin, Cout, n, monkeypatch):
    """Sparse convolutions (SubMConv2d and the strided SparseConv2d, forward and data gradient) in bf16x3 mode on k_gemm_b3f<.., TABLE>
    (conv_gemmf.hip): the tile's slice of the neighbour table staged in LDS, gathered 64-channel chunks, weight fragments from L2.
    Same products in the same order per accumulator as the gathered kernel k_conv_igemm_b3 -> bit-identical to it (RD_SPARSEF = 0 / 1),
    and within 1e-3 of the oracle's pair-list convolution."""
    A, K, SP = _mods()
    rng = np.random.default_rng(Cin + Cout + n)
    B, H, W = 2, 60, 56
    idx = rand_sites(rng, B, H, W, n)
    feats = torch.from_numpy(rng.normal(size=(n, Cin)).astype(np.float32))
    w = torch.from_numpy((rng.normal(size=(Cout, 3, 3, Cin)) / np.sqrt(9 * Cin)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))
    K.set_conv_math("bf16x3")
    try:
        for subm in (True, False):
            res = {}
            for on in (False, True):
                monkeypatch.setattr(K, "SPARSEF", on)
                conv = (SP.SubMConv2d if subm else SP.SparseConv2d)(Cin, Cout, 3, stride=1 if subm else 2, padding=1, bias=True).to(DEV)
                with torch.no_grad():
                    conv.weight.copy_(w); conv.bias.copy_(b)
                fd = feats.to(DEV).requires_grad_(True)
                A.begin_step(torch.device(DEV))
                out = conv(SP.SparseConvTensor(fd, torch.from_numpy(idx).to(DEV), [H, W], B))
                g = np.random.default_rng(7)
                go = torch.from_numpy(g.normal(size=tuple(out.features.shape)).astype(np.float32))
                (out.features * go.to(DEV)).sum().backward()
                torch.cuda.synchronize()
                res[on] = (out.features.detach().clone(), fd.grad.clone(), conv.weight.grad.clone(), go)
            assert torch.equal(res[True][0], res[False][0]), f"forward differs from the gathered kernel (subm={subm})"
            assert torch.equal(res[True][1], res[False][1]), f"data gradient differs from the gathered kernel (subm={subm})"
            fr = feats.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
            nbr = osp.subm_rulebook(idx, (H, W)) if subm else osp.strided_rulebook(idx, (H, W))[2]
            ref = osp.sparse_conv(fr, nbr, wr, br)
            (ref * res[True][3]).sum().backward()
            close(res[True][0], ref, what=f"fwd subm={subm}")
            close(res[True][1], fr.grad, what="dgrad")
            close(res[True][2], wr.grad, atol=2e-4, what="wgrad")
    finally:
        K.set_conv_math("f32")


@pytest.mark.parametrize("Cin,Cout,k,s,p,H,W", [(256, 256, 3, 1, 1, 12, 10), (256, 256, 3, 2, 1, 13, 11), (512, 256, 1, 1, 0, 9, 9),
                                                 (64, 2, 3, 1, 1, 8, 8), (256, 27, 3, 2, 1, 16, 16)])
def test_dense_conv2d_forward_and_backward(Cin, Cout, k, s, p, H, W):
    A, K, SP = _mods()
    rng = np.random.default_rng(Cin * 7 + Cout + k)
    B = 2
    x = torch.from_numpy(rng.normal(size=(B, Cin, H, W)).astype(np.float32))
    conv = torch.nn.Conv2d(Cin, Cout, k, s, p, bias=True)
    xr = x.clone().requires_grad_(True)
    ref = conv(xr)
    go = torch.from_numpy(rng.normal(size=tuple(ref.shape)).astype(np.float32))
    (ref * go).sum().backward()
    import copy
    cd = copy.deepcopy(conv).to(DEV)
    cd.weight.grad = None; cd.bias.grad = None
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rows, _, _, _ = A.nchw_to_rows(xd)
    spec = A.dense_conv_spec(B, H, W, k, k, s, p)
    out = A.rows_to_nchw(A.conv(rows, cd.weight, cd.bias, spec, Cout), B, *spec.out_hw)
    close(out, ref, what="conv2d fwd")
    (out * go.to(DEV)).sum().backward()
    close(xd.grad, xr.grad, what="conv2d dgrad")
    close(cd.weight.grad, conv.weight.grad, atol=2e-4, what="conv2d wgrad")
    close(cd.bias.grad, conv.bias.grad, atol=2e-4, what="conv2d bias grad")


@pytest.mark.parametrize("k,s,p", [(2, 2, 0), (4, 2, 1)])
def test_conv_transpose2d_forward_and_backward(k, s, p):
    A, K, SP = _mods()
    rng = np.random.default_rng(k)
    B, C, H, W = 2, 256, 7, 6
    x = torch.from_numpy(rng.normal(size=(B, C, H, W)).astype(np.float32))
    m = torch.nn.ConvTranspose2d(C, C, k, s, p, bias=True)
    xr = x.clone().requires_grad_(True)
    ref = m(xr)
    go = torch.from_numpy(rng.normal(size=tuple(ref.shape)).astype(np.float32))
    (ref * go).sum().backward()
    import copy
    md = copy.deepcopy(m).to(DEV); md.weight.grad = None; md.bias.grad = None
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rows, _, _, _ = A.nchw_to_rows(xd)
    spec = A.dense_conv_spec(B, H, W, k, k, s, p, transposed=True)
    assert tuple(spec.out_hw) == tuple(ref.shape[2:])
    out = A.rows_to_nchw(A.conv(rows, md.weight, md.bias, spec, C), B, *spec.out_hw)
    close(out, ref, what="convT fwd")
    (out * go.to(DEV)).sum().backward()
    close(xd.grad, xr.grad, what="convT dgrad")
    close(md.weight.grad, m.weight.grad, atol=2e-4, what="convT wgrad")


def test_linear_as_one_tap_conv():
    A, K, SP = _mods()
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.normal(size=(333, 256)).astype(np.float32))
    lin = torch.nn.Linear(256, 1024)
    xr = x.clone().requires_grad_(True); ref = lin(xr)
    go = torch.from_numpy(rng.normal(size=tuple(ref.shape)).astype(np.float32)); (ref * go).sum().backward()
    import copy
    ld = copy.deepcopy(lin).to(DEV); ld.weight.grad = None; ld.bias.grad = None
    xd = x.to(DEV).requires_grad_(True)
    out = A.conv(xd, ld.weight, ld.bias, A.linear_spec(333), 1024)
    close(out, ref); (out * go.to(DEV)).sum().backward()
    close(xd.grad, xr.grad); close(ld.weight.grad, lin.weight.grad, atol=2e-4); close(ld.bias.grad, lin.bias.grad, atol=2e-4)


@pytest.mark.parametrize("rows,Cin,Cout", [(333, 256, 1024),       # ragged rows, 64-row tiles
                                           (40000, 64, 128),       # 128 x 128 tiles, one K chunk, ragged last row tile
                                           (4100, 1024, 256),      # 16 K chunks
                                           (700, 2304, 64),        # DCNv2 column GEMM depth, 64-column tile
                                           (129, 128, 96)])        # Cout not a tile multiple
def test_one_tap_gemm_with_fragment_major_weights(rows, Cin, Cout):
    """k_gemm_b3f (conv_gemmf.hip): nn.Linear / 1x1 convolutions in bf16x3 mode with the weights read as MFMA fragments straight from
    L2.  Same products in the same order per accumulator as the gathered kernel -> equal to it bit for bit (forward, full epilogue,
    statistics to rounding of the atomics), and within 1e-3 of torch's fp32 matmul; forward and data gradient through autograd."""
    A, K, SP = _mods()
    rng = np.random.default_rng(rows + Cin)
    x = torch.from_numpy(rng.normal(size=(rows, Cin)).astype(np.float32))
    w = torch.from_numpy((rng.normal(size=(Cout, Cin)) / np.sqrt(Cin)).astype(np.float32))
    b, sc, sh = [torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32)) for _ in range(3)]
    res = torch.from_numpy(rng.normal(size=(rows, Cout)).astype(np.float32))
    go = torch.from_numpy(rng.normal(size=(rows, Cout)).astype(np.float32))
    pre = x @ w.t() + b
    spec = A.linear_spec(rows)
    K.set_conv_math("bf16x3")
    try:
        assert K.wants_frag_weights(spec.fwd_ix, rows, rows, Cin, Cout, 1)
        xd, wk = x.to(DEV), w.to(DEV).view(Cout, 1, Cin).contiguous()
        w_lds = K.weight_layout_split(wk, Cout, Cin, 1, 0)
        w_frag = K.weight_layout_split(wk, Cout, Cin, 1, 0, frag=True)
        kw = dict(scale=sc.to(DEV), shift=sh.to(DEV), residual=res.to(DEV), relu=True)
        ref = K.conv_fwd(xd, w_lds, 1, b.to(DEV), rows, Cout, spec.fwd_ix, w_split=True, **kw)
        got = K.conv_fwd(xd, w_frag, 1, b.to(DEV), rows, Cout, spec.fwd_ix, w_split=2, **kw)
        assert torch.equal(got, ref)
        close(got, torch.relu(pre * sc + sh + res), rtol=1e-3, atol=1e-4, what="frag gemm + epilogue")
        stats = torch.zeros(2 * Cout, device=DEV)
        plain = K.conv_fwd(xd, w_frag, 1, b.to(DEV), rows, Cout, spec.fwd_ix, stats=stats, w_split=2)
        close(plain, pre, rtol=1e-3, atol=1e-4, what="frag gemm")
        close(stats[:Cout], pre.sum(0), rtol=1e-3, atol=2e-3 * rows ** 0.5); close(stats[Cout:], (pre * pre).sum(0), rtol=1e-3, atol=2e-3 * rows ** 0.5)
        # autograd path: forward and data gradient pick the fragment-major operands by themselves
        xr = x.clone().requires_grad_(True)
        (torch.nn.functional.linear(xr, w, b) * go).sum().backward()
        xg = xd.clone().requires_grad_(True)
        wp, bp = torch.nn.Parameter(w.to(DEV)), torch.nn.Parameter(b.to(DEV))
        A.begin_step(torch.device(DEV))
        y = A.conv(xg, wp, bp, spec, Cout)
        assert torch.equal(y.detach(), plain)
        (y * go.to(DEV)).sum().backward()
        torch.cuda.synchronize()
        close(xg.grad, xr.grad, rtol=1e-3, atol=1e-4, what="frag gemm dgrad")
        close(wp.grad, (go.t() @ x), rtol=1e-3, atol=1e-3 * float((go.t() @ x).abs().max()), what="wgrad next to the frag gemm")
    finally:
        K.set_conv_math("f32")


def test_conv_epilogue_and_fused_stats():
    A, K, SP = _mods()
    rng = np.random.default_rng(21)
    rows, Cin, Cout = 1000, 64, 64
    x = torch.from_numpy(rng.normal(size=(rows, Cin)).astype(np.float32))
    w = torch.from_numpy((rng.normal(size=(Cout, Cin)) / 8).astype(np.float32))
    b, sc, sh = [torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32)) for _ in range(3)]
    res = torch.from_numpy(rng.normal(size=(rows, Cout)).astype(np.float32))
    spec = A.linear_spec(rows)
    out = K.conv_fwd(x.to(DEV), w.to(DEV).view(Cout, 1, Cin), 1, b.to(DEV), rows, Cout, spec.fwd_ix, scale=sc.to(DEV), shift=sh.to(DEV),
                     residual=res.to(DEV), relu=True)
    pre = x @ w.t() + b
    close(out, torch.relu(pre * sc + sh + res))
    stats = torch.zeros(2 * Cout, device=DEV)
    K.conv_fwd(x.to(DEV), w.to(DEV).view(Cout, 1, Cin), 1, b.to(DEV), rows, Cout, spec.fwd_ix, stats=stats)
    close(stats[:Cout], pre.sum(0), atol=2e-4); close(stats[Cout:], (pre * pre).sum(0), atol=2e-4)


@pytest.mark.parametrize("C,rows,act,res", [(32, 5000, 1, True), (256, 777, 1, False), (256, 1000, 2, False), (64, 300, 0, False),
                                            (2688, 600, 1, False), (1100, 90, 1, True)])   # > 512 columns: chunked column reduction
def test_batchnorm_train_forward_backward(C, rows, act, res):
    A, K, SP = _mods()
    rng = np.random.default_rng(C + rows)
    x = torch.from_numpy((rng.normal(size=(rows, C)) * 2 + 0.5).astype(np.float32))
    r = torch.from_numpy(rng.normal(size=(rows, C)).astype(np.float32)) if res else None
    bn = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01)
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)))
        bn.bias.copy_(torch.from_numpy(rng.normal(size=C).astype(np.float32)))
    import copy
    bd = copy.deepcopy(bn).to(DEV)
    xr = x.clone().requires_grad_(True); rr = r.clone().requires_grad_(True) if res else None
    z = bn(xr) + (rr if res else 0)
    ref = torch.relu(z) if act == 1 else (F.gelu(z) if act == 2 else z)
    go = torch.from_numpy(rng.normal(size=(rows, C)).astype(np.float32)); (ref * go).sum().backward()
    xd = x.to(DEV).requires_grad_(True); rd = r.to(DEV).requires_grad_(True) if res else None
    out = A.bn_act_train(xd, bd, rd, act=act)
    close(out, ref); (out * go.to(DEV)).sum().backward()
    close(xd.grad, xr.grad, atol=2e-4); close(bd.weight.grad, bn.weight.grad, atol=2e-4); close(bd.bias.grad, bn.bias.grad, atol=2e-4)
    if res:
        close(rd.grad, rr.grad)
    close(bd.running_mean, bn.running_mean); close(bd.running_var, bn.running_var)
    A.end_forward()            # num_batches_tracked is bumped for all train-mode BatchNorms of a forward in one launch
    assert int(bd.num_batches_tracked) == 1


def _backbone(prefix_radar, grid, seed):
    from radardistill_amd.pcdet.models.backbones_3d import __all__ as REG
    m = REG["Radar_PillarRes18BackBone8x" if prefix_radar else "PillarRes18BackBone8x"](None, 32, np.array([grid, grid, 40]))
    sd = m.state_dict(); seeded_fill_(sd, seed=seed); m.load_state_dict(sd)
    return m.to(DEV)


@pytest.mark.parametrize("training", [False, True])
def test_sparse_enc_c2_vs_oracle(training):
    """BASELINE config C2 at reduced size: radar VFE + SparseEnc (+dense(), conv5): indices bit-exact, features 1e-3."""
    grid, B = 128, 2
    vfe_m, pc_range, voxel, gs = _vfe_module("radar", 6, grid, seed=31)
    bb = _backbone(True, grid, seed=32)
    vfe_m.train(training); bb.train(training)
    batch = make_batch(batch_size=B, n_lidar=16, n_radar=1200, n_boxes=2, grid=grid, seed=11)
    pts = torch.from_numpy(batch["radar_points"])
    bd = {"radar_points": pts.to(DEV), "batch_size": B}
    with torch.set_grad_enabled(training):
        bd = bb(vfe_m(bd))
    ms = bd["radar_multi_scale_2d_features"]
    st = {("radar_vfe." + k): v.detach().cpu().clone() for k, v in vfe_m.state_dict().items()}
    st.update({("radar_backbone_3d." + k): v.detach().cpu().clone() for k, v in bb.state_dict().items()})
    seeded_fill_({k[len("radar_vfe."):]: v for k, v in st.items() if k.startswith("radar_vfe.")}, seed=31)
    seeded_fill_({k[len("radar_backbone_3d."):]: v for k, v in st.items() if k.startswith("radar_backbone_3d.")}, seed=32)
    params = [k for k in st if st[k].is_floating_point() and "running" not in k]
    if training:
        for k in params:
            st[k].requires_grad_(True)
    ov = ovfe.dynamic_pillar_vfe(pts, st, "radar_vfe.", pc_range, voxel, gs, training=training)
    ob = osp.pillar_res18_backbone(ov["pillar_features"], ov["pillar_coords"].numpy(), B, gs, st, "radar_backbone_3d.", training=training)
    assert np.array_equal(bd["radar_pillar_coords"].cpu().numpy(), ov["pillar_coords"].numpy())
    for k in ("x_conv1", "x_conv2", "x_conv3"):
        f, idx, shape = ob[k]
        assert np.array_equal(ms[k].indices.cpu().numpy(), idx), k
        close(ms[k].features, f, what=k)
    close(ms["x_conv4"], ob["x_conv4"], what="x_conv4"); close(ms["x_conv5"], ob["x_conv5"], what="x_conv5")
    if training:
        g4 = torch.from_numpy(np.random.default_rng(1).normal(size=tuple(ob["x_conv4"].shape)).astype(np.float32))
        g5 = torch.from_numpy(np.random.default_rng(2).normal(size=tuple(ob["x_conv5"].shape)).astype(np.float32))
        ((ms["x_conv4"] * g4.to(DEV)).sum() + (ms["x_conv5"] * g5.to(DEV)).sum()).backward()
        ((ob["x_conv4"] * g4).sum() + (ob["x_conv5"] * g5).sum()).backward()
        named = dict(("radar_vfe." + k, p) for k, p in vfe_m.named_parameters())
        named.update(("radar_backbone_3d." + k, p) for k, p in bb.named_parameters())
        worst = 0.0
        gscale = max(float(st[k].grad.abs().max()) for k in params)
        for k in params:
            a, b = named[k].grad.detach().cpu(), st[k].grad
            # Criterion: relative L2 error per parameter tensor.  Every backward OP is held to 1e-3 max-norm in the unit tests above
            # (and RD_DEBUG_CHECK=1 re-derives each op inside the network).  At network level an fp32 forward and the oracle
            # disagree on the SIGN of a handful of ReLU inputs that are ~1e-8 from zero (measured: 1 of 226k at x_conv2,
            # tools/diag/grad_layers.py); each such flip moves a few gradient rows by ~1 % of max, so a max-norm bound of 1e-3 is
            # not meaningful for whole-network gradients.  Conv biases in front of a BatchNorm have a true gradient of exactly 0:
            # absolute floor tied to the global gradient scale.
            err = float((a - b).norm())
            bound = 1e-2 * float(b.norm()) + 1e-4 * gscale * (b.numel() ** 0.5)
            worst = max(worst, err / bound)
            assert err <= bound, (k, err, float(b.norm()), gscale)
        print("worst gradient error / bound", worst)
        # running statistics follow the reference's momentum update
        for k, v in bb.state_dict().items():
            if "running" in k:
                close(v, st["radar_backbone_3d." + k], what=k)


@pytest.mark.parametrize("C,H,W,K", [(256, 16, 16, 7), (64, 9, 11, 7), (32, 5, 5, 3), (96, 37, 33, 7), (36, 8, 8, 7)])
def test_depthwise_conv_forward_and_backward(C, H, W, K):
    """dwconv.hip: 7x7 with C % 32 == 0 takes the LDS-tiled kernels (16x16-pixel tiles: 37x33 has ragged tiles in both directions), other
    shapes (3x3, C = 36) the plain ones."""
    A, K_, SP = _mods()
    rng = np.random.default_rng(C + H)
    B = 2
    x = torch.from_numpy(rng.normal(size=(B, C, H, W)).astype(np.float32))
    conv = torch.nn.Conv2d(C, C, K, padding=K // 2, groups=C)
    xr = x.clone().requires_grad_(True)
    ref = conv(xr)
    go = torch.from_numpy(rng.normal(size=tuple(ref.shape)).astype(np.float32))
    (ref * go).sum().backward()
    import copy
    cd = copy.deepcopy(conv).to(DEV); cd.weight.grad = None; cd.bias.grad = None
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rows, _, _, _ = A.nchw_to_rows(xd)
    out = A.rows_to_nchw(A.dwconv(rows, cd, B, H, W), B, H, W)
    close(out, ref, what="dwconv fwd")
    (out * go.to(DEV)).sum().backward()
    close(xd.grad, xr.grad, what="dwconv dgrad")
    close(cd.weight.grad, conv.weight.grad, atol=2e-4, what="dwconv wgrad")
    close(cd.bias.grad, conv.bias.grad, atol=2e-4, what="dwconv bias grad")


@pytest.mark.parametrize("B,H,W,n_list", [(2, 16, 16, (1, 2, 3, 2, 2, 1, 1)), (1, 13, 10, (3, 4, 1)), (3, 8, 24, (2,))])
def test_batchnorm_relu_narrow_convs_fused_node_vs_torch(B, H, W, n_list):
    """autograd.bn_relu_nconv_train (train-mode BatchNorm + ReLU + the branches' narrow convolutions as one node; backward =
    rd_nconv_dgrad_bn, which recomputes the activation gradient inside the BatchNorm backward's two passes) against torch in fp64:
    output, running statistics, gradients of the BatchNorm input, gamma, beta, conv weights and bias -- and against the unfused two-node
    path of the same library (deterministic mode takes it)."""
    A, K, SP = _mods()
    rng = np.random.default_rng(B * 10 + W)
    NB = len(n_list)
    C = NB * 64
    cols = np.concatenate([[0], np.cumsum(n_list)]).tolist()
    NO = cols[-1]
    raw = torch.from_numpy(rng.normal(0.3, 1.0, size=(B, H, W, C)).astype(np.float32))
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, size=C).astype(np.float32)); beta = torch.from_numpy(rng.normal(0, 0.3, size=C).astype(np.float32))
    w = torch.from_numpy((rng.normal(size=(NO, 64, 3, 3)) * 0.1).astype(np.float32)); b = torch.from_numpy(rng.normal(size=NO).astype(np.float32))
    go = torch.from_numpy(rng.normal(size=(B, H, W, NO)).astype(np.float32))
    rr, gr, br_, wr, cr = [t.double().requires_grad_(True) for t in (raw, gamma, beta, w, b)]
    yr = F.relu(F.batch_norm(rr.permute(0, 3, 1, 2), None, None, gr, br_, True, 0.0, 1e-3))
    ref = torch.cat([F.conv2d(yr[:, 64 * i:64 * i + 64], wr[cols[i]:cols[i + 1]], cr[cols[i]:cols[i + 1]], padding=1) for i in range(NB)], dim=1)
    (ref.permute(0, 2, 3, 1) * go.double()).sum().backward()
    tab = K.BranchTable([64 * i for i in range(NB)], cols[:-1], list(n_list))
    res = []
    for det in (False, True):
        K.set_deterministic(det)
        try:
            rd = raw.reshape(-1, C).to(DEV).requires_grad_(True)
            gd, bd, wd, cd = [t.to(DEV).requires_grad_(True) for t in (gamma, beta, w, b)]
            rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
            A.begin_step(torch.device(DEV))
            out = A.bn_relu_nconv_train(rd, gd, bd, rm, rv, 1e-3, 0.01, None, (), wd, cd, B, H, W, tab)
            (out * go.reshape(-1, NO).to(DEV)).sum().backward()
            A.end_forward()
        finally:
            K.set_deterministic(False)
        close(out, ref.detach().permute(0, 2, 3, 1).reshape(-1, NO), rtol=1e-4, atol=1e-4, what=f"output det={det}")
        close(rd.grad, rr.grad.reshape(-1, C), rtol=1e-3, atol=2e-5, what=f"grad input det={det}")
        close(gd.grad, gr.grad, rtol=1e-3, atol=2e-4, what="grad gamma"); close(bd.grad, br_.grad, rtol=1e-3, atol=2e-4, what="grad beta")
        close(wd.grad, wr.grad, rtol=1e-3, atol=2e-4, what="grad w"); close(cd.grad, cr.grad, rtol=1e-3, atol=2e-4, what="grad bias")
        x2 = raw.reshape(-1, C).double()
        close(rm, 0.01 * x2.mean(0), rtol=1e-4, atol=1e-6, what="running mean"); close(rv, 0.99 + 0.01 * x2.var(0, unbiased=True), rtol=1e-4, atol=1e-6)
        res.append(rd.grad.clone())
    close(res[0], res[1], rtol=1e-4, atol=1e-5, what="fused vs two-node input gradient")


@pytest.mark.parametrize("B,H,W,n_list", [(2, 16, 16, (1, 2, 3, 2, 2, 1, 1)), (1, 13, 10, (3, 4, 1)), (3, 8, 24, (2,))])
def test_narrow_branch_convs_forward_and_backward(B, H, W, n_list):
    """rd_nconv_{fwd,dgrad,wgrad} (all head branches' final 64 -> n convs in one launch) vs torch conv2d per branch."""
    A, K, SP = _mods()
    rng = np.random.default_rng(B * 100 + H)
    NB = len(n_list)
    cols = np.concatenate([[0], np.cumsum(n_list)]).tolist()
    NO = cols[-1]
    y = torch.from_numpy(rng.normal(size=(B, H, W, NB * 64)).astype(np.float32))
    w = torch.from_numpy((rng.normal(size=(NO, 64, 3, 3)) * 0.1).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=NO).astype(np.float32))
    go = torch.from_numpy(rng.normal(size=(B, H, W, NO)).astype(np.float32))
    yr, wr, br = y.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.cat([F.conv2d(yr[..., 64 * i:64 * i + 64].permute(0, 3, 1, 2), wr[cols[i]:cols[i + 1]], br[cols[i]:cols[i + 1]], padding=1)
                     for i in range(NB)], dim=1).permute(0, 2, 3, 1)
    (ref * go).sum().backward()
    tab = K.BranchTable([64 * i for i in range(NB)], cols[:-1], list(n_list))
    yd = y.reshape(-1, NB * 64).to(DEV).requires_grad_(True)
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    A.begin_step(torch.device(DEV))
    out = A.nconv(yd, wd, bd, B, H, W, tab)
    close(out, ref.reshape(-1, NO), what="nconv forward")
    (out * go.reshape(-1, NO).to(DEV)).sum().backward()
    close(yd.grad, yr.grad.reshape(-1, NB * 64), what="nconv dgrad")
    close(wd.grad, wr.grad, what="nconv wgrad")
    close(bd.grad, br.grad, what="nconv bias grad")
    with torch.no_grad():
        close(A.nconv(yd.detach(), wd.detach(), None, B, H, W, tab), (ref - br.detach()).reshape(-1, NO), what="nconv forward, no bias")


def test_narrow_branch_convs_reject_bad_tables():
    A, K, SP = _mods()
    y = torch.zeros((64, 128), device=DEV)
    w = torch.zeros((3, 64, 3, 3), device=DEV)
    tab = K.BranchTable([0, 64], [0, 1], [1, 2])
    with pytest.raises(RuntimeError):
        K.nconv_fwd(y[:, :64].contiguous(), w, None, 1, 8, 8, tab)          # second branch's channels are outside the row
    tab5 = K.BranchTable([0], [0], [5])
    with pytest.raises(RuntimeError):
        K.nconv_fwd(y, torch.zeros((5, 64, 3, 3), device=DEV), None, 1, 8, 8, tab5)   # 5 outputs per branch: not supported


# ------------------------------------------------------------------------------------------ padded-voxel input format (8(f) rank 3)
@pytest.mark.parametrize("B,n_pts,max_points,max_voxels,gz", [(2, 1500, 8, 700, 1), (3, 4000, 5, 100000, 1), (1, 0, 4, 10, 1),
                                                               (2, 3000, 3, 5000, 4)])
def test_hard_voxelizer_bit_exact_vs_oracle(B, n_pts, max_points, max_voxels, gz):
    """rd_voxelize_hard vs the sequential CPU algorithm: voxel order, slot order, truncation by both capacities -- all bit-exact."""
    from oracle import voxel as ovox
    from radardistill_amd.voxel import VoxelGenerator
    pc_range, voxel, grid = bench_geometry(128)
    voxel = [voxel[0], voxel[1], (pc_range[5] - pc_range[2]) / gz]
    batch = make_batch(batch_size=B, n_lidar=n_pts, n_radar=16, n_boxes=2, grid=128, seed=20 + B)
    pts = batch["points"] if n_pts else np.zeros((0, 6), dtype=np.float32)
    if n_pts:
        pts = pts.copy()
        pts[::17, 1] += 40.0                            # some points outside the range
        pts[5::29, 3] = pc_range[5]                      # exactly on the upper z bound: floor lands on grid size -> dropped
    gen = VoxelGenerator(voxel, pc_range, 5, max_points, max_voxels)
    v, c, n = gen.generate(torch.from_numpy(pts).to(DEV), batch_size=B)
    rv, rc, rn = ovox.batch_points_to_voxels(pts, B, voxel, pc_range, max_points, max_voxels)
    assert v.shape == rv.shape and c.dtype == torch.int32 and n.dtype == torch.int32
    assert np.array_equal(c.cpu().numpy(), rc) and np.array_equal(n.cpu().numpy(), rn)
    assert np.array_equal(v.cpu().numpy(), rv)           # copied point words: exact
    if n_pts and max_voxels == 700:
        assert rv.shape[0] == B * max_voxels             # the voxel cap really bites in this case


def test_pillar_vfe_and_scatter_golden(golden_dir):
    """PillarVFE (one fused PFN layer) + PointPillarScatter on HIP vs the reference modules' outputs."""
    from oracle import voxel as ovox
    from radardistill_amd.pcdet.config import AttrDict
    from radardistill_amd.pcdet.models.backbones_3d.vfe import __all__ as VFE
    from radardistill_amd.pcdet.models.backbones_2d.map_to_bev import __all__ as M2B
    from radardistill_amd.voxel import VoxelGenerator
    g = np.load(f"{golden_dir}/g7_pillar.npz")
    pc_range, voxel, grid = bench_geometry(128)
    voxel = [voxel[0], voxel[1], pc_range[5] - pc_range[2]]
    batch = make_batch(batch_size=2, n_lidar=1500, n_radar=16, n_boxes=2, grid=128, seed=8)
    v, c, n = VoxelGenerator(voxel, pc_range, 5, 8, 700).generate(torch.from_numpy(batch["points"]).to(DEV), batch_size=2)
    for tag, use_abs, with_dist in (("a", True, False), ("b", False, True)):
        m = VFE["PillarVFE"](AttrDict(USE_NORM=True, WITH_DISTANCE=with_dist, USE_ABSLOTE_XYZ=use_abs, NUM_FILTERS=[64]), num_point_features=5,
                             voxel_size=voxel, point_cloud_range=pc_range)
        sd = m.state_dict(); seeded_fill_(sd, seed=31); m.load_state_dict(sd); m = m.to(DEV)
        for mode in ("eval", "train"):
            m.train(mode == "train")
            with torch.no_grad():
                bd = m({"voxels": v, "voxel_num_points": n, "voxel_coords": c, "batch_size": 2})
            close(bd["pillar_features"], g[f"{tag}_{mode}_pillar_features"], what=f"{tag} {mode}")
        close(m.pfn_layers[0].norm.running_mean, g[f"{tag}_running_mean"]); close(m.pfn_layers[0].norm.running_var, g[f"{tag}_running_var"])
        if tag == "a":
            m.eval()
            s = M2B["PointPillarScatter"](AttrDict(NUM_BEV_FEATURES=64), grid_size=[int(grid[0]), int(grid[1]), 1])
            with torch.no_grad():
                bd = s(m({"voxels": v, "voxel_num_points": n, "voxel_coords": c, "batch_size": 2}))
            assert bd["spatial_features"].shape == (2, 64, 128, 128)
            close(bd["spatial_features"], g["spatial_features"], what="scatter")


@pytest.mark.parametrize("math", ["f32", "bf16x3"])
def test_pillar_vfe_training_and_multi_layer_golden(golden_dir, math):
    """PillarVFE in training use (pillar_vfe.py:8-49,94-123) on HIP: rd_pillar_decorate -> per PFN layer [1-tap implicit GEMM ->
    BatchNorm rows -> rd_pfn_pool] with HIP backward, vs the reference module's own outputs AND parameter gradients (fixture g9:
    one layer, two layers with the [x | max] concatenation, a 16-channel first layer, USE_NORM False)."""
    A, K, SP = _mods()
    from radardistill_amd.pcdet.config import AttrDict
    from radardistill_amd.pcdet.models.backbones_3d.vfe import __all__ as VFE
    from radardistill_amd.voxel import VoxelGenerator
    from tests.test_oracle_golden import G9_CASES
    g = np.load(f"{golden_dir}/g9_pillar_train.npz")
    pc_range, voxel, grid = bench_geometry(128)
    voxel = [voxel[0], voxel[1], pc_range[5] - pc_range[2]]
    batch = make_batch(batch_size=2, n_lidar=1500, n_radar=16, n_boxes=2, grid=128, seed=8)
    v, c, n = VoxelGenerator(voxel, pc_range, 5, 8, 700).generate(torch.from_numpy(batch["points"]).to(DEV), batch_size=2)
    K.set_conv_math(math)
    try:
        for tag, use_abs, with_dist, use_norm, filters in G9_CASES:
            m = VFE["PillarVFE"](AttrDict(USE_NORM=use_norm, WITH_DISTANCE=with_dist, USE_ABSLOTE_XYZ=use_abs, NUM_FILTERS=filters),
                                 num_point_features=5, voxel_size=voxel, point_cloud_range=pc_range)
            sd = m.state_dict(); seeded_fill_(sd, seed=41); m.load_state_dict(sd); m = m.to(DEV)
            m.train()
            A.begin_step(torch.device(DEV))
            feats = m({"voxels": v, "voxel_num_points": n, "voxel_coords": c, "batch_size": 2})["pillar_features"]
            A.end_forward()
            go = torch.from_numpy(np.random.default_rng(9).normal(size=tuple(feats.shape)).astype(np.float32)).to(DEV)
            (feats * go).sum().backward()
            close(feats, g[f"{tag}_features"], what=f"{tag} train features")
            for k, p in m.named_parameters():
                ref = g[f"{tag}_grad_{k}"]
                close(p.grad, ref, rtol=2e-3, atol=2e-5 * float(np.abs(ref).max()) + 1e-6, what=f"{tag} grad {k}")
            for k, b in m.named_buffers():
                if "running" in k:
                    close(b, g[f"{tag}_{k}"], what=f"{tag} {k}")
                elif k.endswith("num_batches_tracked"):
                    assert int(b) == 1
            m.eval()
            with torch.no_grad():
                ev = m({"voxels": v, "voxel_num_points": n, "voxel_coords": c, "batch_size": 2})["pillar_features"]
            close(ev, g[f"{tag}_eval_features"], what=f"{tag} eval features")
    finally:
        K.set_conv_math("f32")


def test_full_size_voxelizer_and_rulebook_pyramid_bit_exact():
    """BASELINE configs[1]/[3] at their real size (B = 8, 512 x 512 pillars, 35 k LiDAR points per sample): pillar rows of the HIP
    voxeliser and all four levels of neighbour tables vs the numpy oracle, bit-exact (the oracle's index work is vectorised numpy,
    seconds at this size)."""
    A, K, SP = _mods()
    pc_range, voxel, gs = bench_geometry(512)
    batch = make_batch(batch_size=8, n_lidar=35000, n_radar=2000, n_boxes=30, grid=512, seed=1)
    pts = torch.from_numpy(batch["points"])
    gx, gy = int(gs[0]), int(gs[1])
    _, _, unq, inv, _ = ovfe.voxelize(pts, pc_range, voxel, (gx, gy))
    unq = unq.numpy().astype(np.int64)
    cr = np.stack([unq // (gx * gy), unq % gy, (unq // gy) % gx], axis=1).astype(np.int32)          # key = (b, cx, cy) -> row (b, y, x)
    rg, point_row = K.voxelize(pts.to(DEV), 8, gx, gy, pc_range[0], pc_range[1], voxel[0], voxel[1])
    n = int(K.rankgrid_count_tensor(rg, 8 * gx * gy).item())
    coords = K.rankgrid_coords(rg, 8, gy, gx, True, n)
    assert n == cr.shape[0] and np.array_equal(coords.cpu().numpy(), cr)
    pr = point_row.cpu().numpy()
    assert int((pr >= 0).sum()) == inv.shape[0] and np.array_equal(pr[pr >= 0], inv.numpy())
    SP.register_rankgrid(coords, rg, True)
    t = SP.SparseConvTensor(torch.zeros((n, 32), device=DEV), coords, [int(gs[1]), int(gs[0])], 8)
    idx = cr
    H, W = int(gs[1]), int(gs[0])
    for level in range(4):
        nbr = t._level.subm_spec().fwd_nbr.cpu().numpy()
        assert np.array_equal(nbr, osp.subm_rulebook(idx, (H, W))), f"SubM table, level {level}"
        # symmetry of a sub-manifold rulebook: i is tap t of j  <=>  j is tap 8 - t of i
        o, tt = np.nonzero(nbr >= 0)
        assert np.array_equal(nbr[nbr[o, tt], 8 - tt], o)
        if level == 3:
            break
        lvl, spec = t._level.down()
        oidx, oshape, snbr = osp.strided_rulebook(idx, (H, W))
        assert np.array_equal(lvl.coords.cpu().numpy(), oidx) and np.array_equal(spec.fwd_nbr.cpu().numpy(), snbr), f"strided table, level {level}"
        t = SP.SparseConvTensor(torch.zeros((oidx.shape[0], 32), device=DEV), lvl.coords, [lvl.H, lvl.W], 8, _level=lvl)
        idx, (H, W) = oidx, oshape


@pytest.mark.parametrize("B,grid,n_lidar", [(8, 512, 35000), (2, 96, 600), (1, 40, 0)])
def test_geometry_prelude_calls_equal_the_separate_entry_points(B, grid, n_lidar):
    """rd_geometry_begin / rd_geometry_finish (two calls per branch around the step's one read) fill, word for word, the rank grids,
    point rows, sizes, coordinates and the 4 + 3 + 3 neighbour tables that the separate entry points do; at the bench size, on a ragged
    small grid, and for a sample without any point (every level empty)."""
    A, K, SP = _mods()
    pc_range, voxel, gs = bench_geometry(grid)
    gx, gy = int(gs[0]), int(gs[1])
    if n_lidar:
        pts = torch.from_numpy(make_batch(batch_size=B, n_lidar=n_lidar, n_radar=50, n_boxes=5, grid=grid, seed=4)["points"]).to(DEV)
    else:
        pts = torch.zeros((0, 6), device=DEV)
    rg, point_row = K.voxelize(pts, B, gx, gy, pc_range[0], pc_range[1], voxel[0], voxel[1])
    P = int(K.rankgrid_count_tensor(rg, B * gx * gy))
    marked = SP.mark_pyramid(rg, True, B, gy, gx, 3)
    rows = [P] + [int(m[3]) for m in marked]
    level = SP._Level(K.rankgrid_coords(rg, B, gy, gx, True, P), rg, True, B, gy, gx)
    SP.finish_pyramid(level, marked, rows[1:])

    scal = torch.full((5,), -7, dtype=torch.int32, device=DEV)
    rgs, point_row2, dims = K.geometry_begin(pts, B, gx, gy, pc_range[0], pc_range[1], voxel[0], voxel[1], 3, scal)
    assert scal.tolist() == [P, int((point_row >= 0).sum()), *rows[1:]]
    assert torch.equal(point_row, point_row2)
    for a, b in zip([rg] + [m[0] for m in marked], rgs):
        assert torch.equal(a, b)
    coords, subm, down, up = K.geometry_finish(rgs, B, gx, gy, rows)
    top = SP.pyramid_from_tables(rgs, dims, B, coords, subm, down, up)
    ref, got = level, top
    for l in range(4):
        assert (ref.n, ref.H, ref.W, ref.xmajor) == (got.n, got.H, got.W, got.xmajor) and ref.n == rows[l]
        assert torch.equal(ref.coords, got.coords) and torch.equal(ref._subm.fwd_nbr, got._subm.fwd_nbr)
        assert (got._subm.fwd_ix.flip, got._subm.bwd_ix.flip) == (ref._subm.fwd_ix.flip, ref._subm.bwd_ix.flip)
        if l == 3:
            assert got._down is None
            break
        assert torch.equal(ref._down[1].fwd_nbr, got._down[1].fwd_nbr) and torch.equal(ref._down[1].bwd_nbr, got._down[1].bwd_nbr)
        assert (got._down[1].in_rows, got._down[1].out_rows) == (ref._down[1].in_rows, ref._down[1].out_rows)
        ref, got = ref._down[0], got._down[0]


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 21, 19, 64, 96),       # ragged map, one 64-channel ci tile, Cout not a tile multiple
                                            (8, 64, 64, 256, 256),     # the DenseEnc shape of the bench (512 pixel tiles)
                                            (1, 8, 8, 32, 64),         # a single tile, half-empty ci tile
                                            (3, 16, 24, 512, 64),      # 8 ci tiles, half-empty co tile (CenterHead shared conv)
                                            (2, 32, 32, 64, 448)])     # batched head branches: Cout not a multiple of 128
def test_halo_wgrad_3x3_bf16x3(B, H, W, Cin, Cout):
    """k_conv_wgrad_d3_b3 (conv_wgrad_d3.hip: dense stride-1 3x3 weight gradient, all nine taps on one staged tile, transposing LDS
    reads) against torch's conv2d weight gradient in fp64 at the bf16x3 bound (1e-3 of max; observed ~1e-5), in the atomic
    (chunked) and in the deterministic (one chunk) launch shape; the two must also agree with the gathered kernel it replaces."""
    import os
    from radardistill_amd import autograd as A, kernels as K
    g = np.random.default_rng(B * 100 + H + Cin)
    x = torch.from_numpy(g.normal(size=(B, Cin, H, W)).astype(np.float32))
    go = torch.from_numpy(g.normal(size=(B, Cout, H, W)).astype(np.float32))
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), w, None, 1, 1) * go.double()).sum().backward()
    ref = w.grad.permute(0, 2, 3, 1).reshape(Cout, 9, Cin)                      # kernel layout [Cout][tap][Cin]
    xr = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().to(DEV)
    gr = go.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().to(DEV)
    spec = A.dense_conv_spec(B, H, W, 3, 3, 1, 1)
    K.set_conv_math("bf16x3")
    try:
        A.begin_step(torch.device(DEV))
        got = K.conv_wgrad(xr, gr, 9, spec.fwd_ix).clone()
        K.set_deterministic(True)
        det = K.conv_wgrad(xr, gr, 9, spec.fwd_ix).clone()
        det2 = K.conv_wgrad(xr, gr, 9, spec.fwd_ix).clone()
        K.set_deterministic(False)
    finally:
        K.set_deterministic(False)
        K.set_conv_math("f32")
    scale = float(ref.abs().max())
    for name, t in (("chunked", got), ("one chunk", det)):
        err = float((t.cpu().double() - ref).abs().max())
        assert err <= 1e-3 * scale, (name, err, scale)
    assert torch.equal(det, det2)
    print("halo wgrad max err / max", float((got.cpu().double() - ref).abs().max()) / scale)


@pytest.mark.parametrize("B,H,W,Cin,Cout,kernel", [(2, 64, 64, 64, 256, "8x16x64"),      # 256 workgroups of the 8x16-pixel x 64-channel tile
                                                   (3, 61, 70, 64, 160, "8x16x64"),      # ragged map, half-empty column tile
                                                   (8, 64, 64, 32, 256, "8x16x128"),     # >= 384 big tiles
                                                   (1, 20, 21, 64, 96, "8x8x64")])       # small map
def test_halo_conv_3x3_tile_variants_bf16x3(B, H, W, Cin, Cout, kernel):
    """k_conv_d3_b3 (dense stride-1 3x3, bf16x3 mode) in each of the tile shapes its launcher picks -- `kernel` names the one these
    sizes select (launch_conv_d3_b3: 8x16x128 from 384 big tiles, 8x16x64 while that gives 256 workgroups, else 8x8x64) -- forward
    with fused statistics, data gradient (mirrored taps) and weight gradient against torch in fp64."""
    A, K, SP = _mods()
    big_rows = B * ((H + 7) // 8) * ((W + 15) // 16)
    picked = "8x16x128" if big_rows * ((Cout + 127) // 128) >= 384 else ("8x16x64" if big_rows * ((Cout + 63) // 64) >= 256 else "8x8x64")
    assert picked == kernel
    g = np.random.default_rng(H * 7 + Cout)
    x = torch.from_numpy(g.normal(size=(B, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy((g.normal(size=(Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32))
    b = torch.from_numpy(g.normal(size=Cout).astype(np.float32))
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, 1, 1)
    go = torch.from_numpy(g.normal(size=tuple(ref.shape)).astype(np.float32))
    (ref * go.double()).sum().backward()
    ref_rows = ref.detach().permute(0, 2, 3, 1).reshape(-1, Cout)
    K.set_conv_math("bf16x3")
    try:
        spec = A.dense_conv_spec(B, H, W, 3, 3, 1, 1)
        xd = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().to(DEV).requires_grad_(True)
        wd, bd = torch.nn.Parameter(w.to(DEV)), torch.nn.Parameter(b.to(DEV))
        stats = torch.zeros(2 * Cout, device=DEV)
        A.begin_step(torch.device(DEV))
        out = A.conv(xd, wd, bd, spec, Cout, stats)
        close(out, ref_rows, rtol=1e-4, atol=1e-4, what="halo conv forward")
        close(stats[:Cout], ref_rows.sum(0), rtol=1e-4, atol=1e-3); close(stats[Cout:], (ref_rows * ref_rows).sum(0), rtol=1e-4, atol=1e-3)
        (out * go.permute(0, 2, 3, 1).reshape(-1, Cout).to(DEV)).sum().backward()
        close(xd.grad, xr.grad.permute(0, 2, 3, 1).reshape(-1, Cin), rtol=1e-4, atol=1e-4, what="halo conv data gradient")
        close(wd.grad, wr.grad, rtol=1e-4, atol=1e-3, what="weight gradient"); close(bd.grad, br.grad, rtol=1e-4, atol=1e-3, what="bias gradient")
    finally:
        K.set_conv_math("f32")


@pytest.mark.parametrize("rows,C", [(8 * 32 * 32, 256), (37, 64), (5, 1024), (1, 256)])
def test_layernorm_rows_vs_torch(rows, C):
    """layernorm.hip (one wavefront per channels-last row, shuffle reductions) against F.layer_norm on the CPU: output and the three
    gradients at the fp32 bound (1e-5 relative: both are fp32, only the summation order differs)."""
    from radardistill_amd import autograd as A
    g = np.random.default_rng(rows + C)
    x = torch.from_numpy(g.normal(1.0, 2.0, size=(rows, C)).astype(np.float32))
    w = torch.from_numpy(g.uniform(0.5, 1.5, size=C).astype(np.float32))
    b = torch.from_numpy(g.normal(size=C).astype(np.float32))
    go = torch.from_numpy(g.normal(size=(rows, C)).astype(np.float32))
    xr, wr, br = [t.clone().requires_grad_(True) for t in (x, w, b)]
    ref = F.layer_norm(xr, (C,), wr, br, 1e-6)
    (ref * go).sum().backward()
    xd, wd, bd = [t.to(DEV).requires_grad_(True) for t in (x, w, b)]
    out = A.layer_norm_rows(xd, wd, bd, 1e-6)
    (out * go.to(DEV)).sum().backward()
    close(out, ref, rtol=1e-5, atol=1e-5, what="layernorm fwd")
    close(xd.grad, xr.grad, rtol=1e-4, atol=1e-5, what="layernorm grad x")
    close(wd.grad, wr.grad, rtol=1e-4, atol=1e-5, what="layernorm grad gamma")
    close(bd.grad, br.grad, rtol=1e-4, atol=1e-5, what="layernorm grad beta")


def test_cat_channels_vs_torch_cat():
    """rd_cat2_rows / rd_split2_rows (torch.cat((a, b), dim=1) of two channels-last maps and its backward): bit-equal values and
    gradients, and the gradients come back contiguous."""
    from radardistill_amd import autograd as A
    g = torch.Generator(device="cpu").manual_seed(2)
    a = torch.randn(2, 12, 9, 7, generator=g).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = torch.randn(2, 8, 9, 7, generator=g).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    a2, b2 = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    out = A.cat_channels(a, b)
    ref = torch.cat((a2, b2), dim=1)
    assert out.shape == ref.shape and torch.equal(out, ref)
    go = torch.randn(ref.shape, generator=g).to(DEV)
    (out * go).sum().backward()
    (ref * go).sum().backward()
    assert torch.equal(a.grad, a2.grad) and torch.equal(b.grad, b2.grad)
    with torch.no_grad():
        assert torch.equal(A.cat_channels(a.detach(), b.detach()), ref.detach())


@pytest.mark.parametrize("rows", [1, 33, 700, 4099])
def test_small_sparse_conv_32_channels_epilogue_and_stats(rows):
    """conv_small.hip (one wavefront per 32 output rows, fragments straight from global memory; bf16x3 mode, 32 -> 32 channels,
    neighbour-table geometry): the full epilogue contract (bias, folded BatchNorm scale / shift, residual, ReLU), the fused column
    statistics, the mirrored table of the data gradient (flip) and ragged row counts, against a float64 gather-and-matmul."""
    A, K, SP = _mods()
    K.set_conv_math("bf16x3")
    try:
        _small_sparse_case(K, rows)
    finally:
        K.set_conv_math("f32")          # the library's arithmetic mode is process-global: leave it as the other tests expect it


def _small_sparse_case(K, rows):
    rng = np.random.default_rng(rows)
    n_in = rows + 5
    nbr = rng.integers(-1, n_in, size=(rows, 9)).astype(np.int32)
    nbr[rng.random((rows, 9)) < 0.4] = -1
    x = rng.normal(size=(n_in, 32)).astype(np.float32)
    w = (rng.normal(size=(32, 9, 32)) / np.sqrt(288)).astype(np.float32)
    b, sc, sh = [rng.normal(size=32).astype(np.float32) for _ in range(3)]
    res = rng.normal(size=(rows, 32)).astype(np.float32)
    nbr_d = torch.from_numpy(nbr).to(DEV)
    for flip in (False, True):
        ix = K.conv_index_table(nbr_d, flip=flip)
        pre = np.zeros((rows, 32))
        for t in range(9):
            src = nbr[:, 8 - t] if flip else nbr[:, t]
            ok = src >= 0
            pre[ok] += x[src[ok]].astype(np.float64) @ w[:, t, :].astype(np.float64).T
        pre += b
        out = K.conv_fwd(torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV), 9, torch.from_numpy(b).to(DEV), rows, 32, ix,
                         scale=torch.from_numpy(sc).to(DEV), shift=torch.from_numpy(sh).to(DEV), residual=torch.from_numpy(res).to(DEV),
                         relu=True, nbr_keepalive=nbr_d)
        close(out, np.maximum(pre * sc + sh + res, 0.0), rtol=1e-4, atol=1e-5, what=f"epilogue flip={flip}")
        stats = torch.zeros(64, device=DEV)
        raw = K.conv_fwd(torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV), 9, torch.from_numpy(b).to(DEV), rows, 32, ix, stats=stats,
                         nbr_keepalive=nbr_d)
        close(raw, pre, rtol=1e-4, atol=1e-5, what="raw")
        close(stats[:32], pre.sum(0), rtol=1e-4, atol=2e-4, what="column sums")
        close(stats[32:], (pre * pre).sum(0), rtol=1e-4, atol=2e-4, what="column sums of squares")


@pytest.mark.parametrize("case", ["sparse", "linear", "strided", "transposed"])
def test_gathered_wgrad_bf16x3_transposing_reads(case):
    """k_conv_wgrad_tr_b3 (bf16x3 mode, Cin / Cout >= 64: row-major LDS images + ds_read_b64_tr_b16 fragments) on ragged shapes --
    rows not a multiple of the 32-row K step, Cout not a multiple of the 128 tile, neighbour tables with holes, stride-2 and
    transposed dense geometry -- against float64 autograd of the same convolution."""
    A, K, SP = _mods()
    rng = np.random.default_rng(len(case))
    K.set_conv_math("bf16x3")
    try:
        if case == "sparse":
            rows, n_in, Cin, Cout = 1237, 1300, 64, 96
            nbr = rng.integers(-1, n_in, size=(rows, 9)).astype(np.int32)
            nbr[rng.random((rows, 9)) < 0.5] = -1
            x = rng.normal(size=(n_in, Cin)).astype(np.float32)
            go = rng.normal(size=(rows, Cout)).astype(np.float32)
            nbr_d = torch.from_numpy(nbr).to(DEV)
            gw = K.conv_wgrad(torch.from_numpy(x).to(DEV), torch.from_numpy(go).to(DEV), 9, K.conv_index_table(nbr_d), nbr_keepalive=nbr_d)
            ref = np.zeros((Cout, 9, Cin))
            for t in range(9):
                ok = nbr[:, t] >= 0
                ref[:, t, :] = go[ok].astype(np.float64).T @ x[nbr[ok, t]].astype(np.float64)
            close(gw, ref, rtol=1e-4, atol=1e-4, what="sparse wgrad")
            return
        if case == "linear":
            B, H, W, Cin, Cout, k, s, p, tr = 1, 1000, 1, 128, 192, 1, 1, 0, False
        elif case == "strided":
            B, H, W, Cin, Cout, k, s, p, tr = 2, 13, 11, 64, 160, 3, 2, 1, False
        else:
            B, H, W, Cin, Cout, k, s, p, tr = 2, 9, 7, 128, 64, 4, 2, 1, True
        x = torch.from_numpy(rng.normal(size=(B, Cin, H, W))).double()
        w = torch.from_numpy(rng.normal(size=(Cin, Cout, k, k) if tr else (Cout, Cin, k, k))).double().requires_grad_(True)
        y = F.conv_transpose2d(x, w, None, s, p) if tr else F.conv2d(x, w, None, s, p)
        go = torch.from_numpy(rng.normal(size=tuple(y.shape)))
        (y * go).sum().backward()
        spec = A.dense_conv_spec(B, H, W, k, k, s, p, transposed=tr)
        xr = x.permute(0, 2, 3, 1).reshape(-1, Cin).float().contiguous().to(DEV)
        gr = go.permute(0, 2, 3, 1).reshape(-1, Cout).float().contiguous().to(DEV)
        gwk = K.conv_wgrad(xr, gr, k * k, spec.fwd_ix)                                   # kernel layout [Cout][taps][Cin]
        ref = (w.grad.permute(1, 2, 3, 0) if tr else w.grad.permute(0, 2, 3, 1)).reshape(Cout, k * k, Cin)
        close(gwk, ref, rtol=1e-4, atol=1e-4, what=f"{case} wgrad")
    finally:
        K.set_conv_math("f32")


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 9, 7, 64, 128), (1, 32, 32, 256, 256), (3, 5, 6, 128, 64)])
def test_conv_transpose_4x4_s2_subpixel_form_bf16x3(B, H, W, Cin, Cout):
    """ConvTranspose2d(k 4, s 2, p 1) in bf16x3 mode runs in its sub-pixel form (k_conv_igemm_b3<.., 11>: one tile = input pixels of one
    output parity class, 4 taps, rows scattered to the class's output pixels): forward incl. fused statistics, data and weight
    gradients against torch on the CPU."""
    A, K, SP = _mods()
    g = np.random.default_rng(B * 100 + H)
    x = torch.from_numpy(g.normal(size=(B, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy((g.normal(size=(Cin, Cout, 4, 4)) / np.sqrt(4 * Cin)).astype(np.float32))
    b = torch.from_numpy(g.normal(size=Cout).astype(np.float32))
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv_transpose2d(xr, wr, br, 2, 1)
    go = torch.from_numpy(g.normal(size=tuple(ref.shape)).astype(np.float32))
    (ref * go).sum().backward()
    K.set_conv_math("bf16x3")
    try:
        spec = A.dense_conv_spec(B, H, W, 4, 4, 2, 1, transposed=True)
        xd = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().to(DEV).requires_grad_(True)
        wd, bd = torch.nn.Parameter(w.to(DEV)), torch.nn.Parameter(b.to(DEV))
        stats = torch.zeros(2 * Cout, device=DEV)
        A.begin_step(torch.device(DEV))
        out = A.conv(xd, wd, bd, spec, Cout, stats)
        ref_rows = ref.detach().permute(0, 2, 3, 1).reshape(-1, Cout)
        close(out, ref_rows, rtol=1e-4, atol=1e-4, what="sub-pixel transposed conv")
        close(stats[:Cout], ref_rows.sum(0), rtol=1e-4, atol=2e-4); close(stats[Cout:], (ref_rows * ref_rows).sum(0), rtol=1e-4, atol=2e-4)
        (out * go.permute(0, 2, 3, 1).reshape(-1, Cout).to(DEV)).sum().backward()
        close(xd.grad, xr.grad.permute(0, 2, 3, 1).reshape(-1, Cin), rtol=1e-4, atol=1e-4, what="dgrad")
        close(wd.grad, wr.grad, rtol=1e-4, atol=2e-4, what="wgrad"); close(bd.grad, br.grad, rtol=1e-4, atol=2e-4, what="bias grad")
    finally:
        K.set_conv_math("f32")


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,p", [(2, 13, 11, 64, 128, 3, 1), (1, 16, 16, 128, 64, 3, 1), (2, 7, 9, 64, 64, 1, 0)])
def test_stride2_conv_data_gradient_subpixel_form_bf16x3(B, H, W, Cin, Cout, k, p):
    """Data gradient of a stride-2 convolution in bf16x3 mode = stride-2 transposed geometry in the sub-pixel form: odd map sizes
    (output pixels whose parity class has no partner), 3x3 (1 / 2 / 2 / 4 taps per class) and 1x1 (one class owns the only tap)."""
    A, K, SP = _mods()
    g = np.random.default_rng(H * 31 + W)
    x = torch.from_numpy(g.normal(size=(B, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy((g.normal(size=(Cout, Cin, k, k)) / np.sqrt(k * k * Cin)).astype(np.float32))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, 2, p)
    go = torch.from_numpy(g.normal(size=tuple(ref.shape)).astype(np.float32))
    (ref * go).sum().backward()
    K.set_conv_math("bf16x3")
    try:
        spec = A.dense_conv_spec(B, H, W, k, k, 2, p)
        xd = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().to(DEV).requires_grad_(True)
        wd = torch.nn.Parameter(w.to(DEV))
        A.begin_step(torch.device(DEV))
        out = A.conv(xd, wd, None, spec, Cout)
        close(out, ref.detach().permute(0, 2, 3, 1).reshape(-1, Cout), rtol=1e-4, atol=1e-4, what="strided conv")
        (out * go.permute(0, 2, 3, 1).reshape(-1, Cout).to(DEV)).sum().backward()
        close(xd.grad, xr.grad.permute(0, 2, 3, 1).reshape(-1, Cin), rtol=1e-4, atol=1e-4, what="dgrad (sub-pixel form)")
        close(wd.grad, wr.grad, rtol=1e-4, atol=2e-4, what="wgrad")
    finally:
        K.set_conv_math("f32")


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride", [(2, 20, 24, 256, 27, 2), (1, 13, 11, 96, 27, 2), (2, 9, 16, 64, 5, 1), (1, 32, 32, 256, 32, 2)])
def test_narrow_output_conv_split_k_bf16x3(B, H, W, Cin, Cout, stride):
    """conv_small.hip k_conv_narrow_b3 (the CMA blocks' 256 -> 27 DCNv2 offset / mask convolution: one wavefront per 32 output pixels,
    contraction split over 32-channel slices combined with atomics into a zero-filled output, bias added by slice 0): forward through
    the autograd layer against torch on the CPU -- stride 2 and 1, ragged maps (rows not a multiple of 32 / 128), 3 and 8 slices --
    plus the gradients of the same layer (unchanged kernels) and the deterministic mode, which keeps the tiled kernel."""
    A, K, SP = _mods()
    g = np.random.default_rng(B * 1000 + H * 10 + Cin)
    x = torch.from_numpy(g.normal(size=(B, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy((g.normal(size=(Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32))
    b = torch.from_numpy(g.normal(size=Cout).astype(np.float32))
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(xr.double(), wr.double(), br.double(), stride=stride, padding=1)
    go = torch.from_numpy(g.normal(size=tuple(ref.shape)).astype(np.float32))
    (ref * go.double()).sum().backward()
    ref_rows = ref.detach().permute(0, 2, 3, 1).reshape(-1, Cout)
    K.set_conv_math("bf16x3")
    try:
        spec = A.dense_conv_spec(B, H, W, 3, 3, stride, 1)
        for det in (False, True):
            K.set_deterministic(det)
            xd = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().to(DEV).requires_grad_(True)
            wd, bd = torch.nn.Parameter(w.to(DEV)), torch.nn.Parameter(b.to(DEV))
            A.begin_step(torch.device(DEV))
            out = A.conv(xd, wd, bd, spec, Cout, None)
            close(out, ref_rows, rtol=1e-4, atol=1e-4, what=f"narrow conv deterministic={det}")
            (out * go.permute(0, 2, 3, 1).reshape(-1, Cout).to(DEV)).sum().backward()
            close(xd.grad, xr.grad.permute(0, 2, 3, 1).reshape(-1, Cin), rtol=1e-4, atol=1e-4, what="dgrad")
            close(wd.grad, wr.grad, rtol=1e-4, atol=2e-4, what="wgrad"); close(bd.grad, br.grad, rtol=1e-4, atol=2e-4, what="bias grad")
    finally:
        K.set_deterministic(False)
        K.set_conv_math("f32")
