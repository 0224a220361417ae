"""CPU known-answer tests that pin the parts of the oracle no reference fixture covers:
spconv semantics (sparse == dense conv sampled at the output sites), DCNv2 identities restated from the
reference's pcdet/ops/basicblock/test.py:69-110,405-435, rotated overlap analytic cases, optimizer."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import bev, head, optim, sparse

torch.set_num_threads(4)


def _rand_sites(rng, B, H, W, n):
    keys = rng.choice(B * H * W, size=n, replace=False)
    keys.sort()
    return np.stack([keys // (H * W), (keys // W) % H, keys % W], axis=1).astype(np.int32)


def test_subm_equals_dense_conv_at_active_sites():
    rng = np.random.default_rng(0)
    B, H, W, Cin, Cout = 2, 12, 10, 5, 7
    idx = _rand_sites(rng, B, H, W, 60)
    feats = torch.from_numpy(rng.normal(size=(60, Cin)).astype(np.float32))
    w = torch.from_numpy(rng.normal(size=(Cout, 3, 3, Cin)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=(Cout,)).astype(np.float32))
    nbr = sparse.subm_rulebook(idx, (H, W))
    out = sparse.sparse_conv(feats, nbr, w, b)
    dense = sparse.to_dense(feats, idx, B, (H, W))
    ref = F.conv2d(dense, w.permute(0, 3, 1, 2), b, padding=1)          # [Cout,kh,kw,Cin] -> [Cout,Cin,kh,kw]
    ref_rows = ref.permute(0, 2, 3, 1)[idx[:, 0], idx[:, 1], idx[:, 2]]
    np.testing.assert_allclose(out.numpy(), ref_rows.numpy(), rtol=1e-5, atol=1e-5)
    # SubM rulebook symmetry: j is tap t of i  <=>  i is tap 8-t of j
    for t in range(9):
        o = np.nonzero(nbr[:, t] >= 0)[0]
        assert np.array_equal(nbr[nbr[o, t], 8 - t], o)
    assert np.array_equal(nbr[:, 4], np.arange(60))


@pytest.mark.parametrize("H,W", [(12, 10), (13, 9)])
def test_strided_equals_dense_conv(H, W):
    rng = np.random.default_rng(1)
    B, Cin, Cout = 2, 4, 6
    idx = _rand_sites(rng, B, H, W, 40)
    feats = torch.from_numpy(rng.normal(size=(40, Cin)).astype(np.float32))
    w = torch.from_numpy(rng.normal(size=(Cout, 3, 3, Cin)).astype(np.float32))
    oidx, oshape, nbr = sparse.strided_rulebook(idx, (H, W))
    out = sparse.sparse_conv(feats, nbr, w)
    dense = sparse.to_dense(feats, idx, B, (H, W))
    ref = F.conv2d(dense, w.permute(0, 3, 1, 2), None, stride=2, padding=1)
    assert tuple(ref.shape[2:]) == tuple(oshape)
    # active outputs == cells whose receptive field holds an active input
    occ = F.conv2d((dense.abs().sum(1, keepdim=True) > 0).float(), torch.ones(1, 1, 3, 3), stride=2, padding=1) > 0
    exp_idx = torch.nonzero(occ[:, 0]).numpy().astype(np.int32)           # sorted (b, y, x)
    assert np.array_equal(oidx, exp_idx)
    ref_rows = ref.permute(0, 2, 3, 1)[oidx[:, 0], oidx[:, 1], oidx[:, 2]]
    np.testing.assert_allclose(out.numpy(), ref_rows.numpy(), rtol=1e-5, atol=1e-5)


def test_empty_sparse_input():
    idx = np.zeros((0, 3), np.int32)
    nbr = sparse.subm_rulebook(idx, (8, 8))
    assert nbr.shape == (0, 9)
    oidx, oshape, snbr = sparse.strided_rulebook(idx, (8, 8))
    assert oidx.shape == (0, 3) and oshape == (4, 4) and snbr.shape == (0, 9)


def test_dcn_zero_offset_unit_mask_is_conv2d():
    """pcdet/ops/basicblock/test.py:69-110 (check_mdconv_zero_offset)."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 9, 8, generator=g)
    w = torch.randn(6, 4, 3, 3, generator=g)
    b = torch.randn(6, generator=g)
    for stride in (1, 2):
        Ho = (9 + 2 - 3) // stride + 1
        Wo = (8 + 2 - 3) // stride + 1
        off = torch.zeros(2, 18, Ho, Wo)
        m = torch.ones(2, 9, Ho, Wo)
        out = bev.modulated_deform_conv(x, off, m, w, b, stride=stride, pad=1)
        ref = F.conv2d(x, w, b, stride=stride, padding=1)
        assert (out - ref).abs().max() < 1e-5


def test_dcn_integer_offset_shifts_sampling():
    x = torch.arange(2 * 1 * 6 * 6, dtype=torch.float32).view(2, 1, 6, 6)
    w = torch.zeros(1, 1, 3, 3); w[0, 0, 1, 1] = 1.0                      # centre tap only
    off = torch.zeros(2, 18, 6, 6); off[:, 2 * 4] = 1.0; off[:, 2 * 4 + 1] = -2.0   # tap 4: dh=+1, dw=-2
    out = bev.modulated_deform_conv(x, off, torch.ones(2, 9, 6, 6), w, torch.zeros(1), stride=1, pad=1)
    exp = torch.zeros_like(x)
    exp[:, :, :5, 2:] = x[:, :, 1:, :4]
    assert torch.equal(out, exp)


def test_dcn_gradcheck_fp64():
    """pcdet/ops/basicblock/test.py:405-435 (gradcheck eps 1e-3, atol 1e-3, rtol 1e-2)."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 2, 5, 5, generator=g, dtype=torch.float64, requires_grad=True)
    off = (torch.randn(1, 18, 3, 3, generator=g, dtype=torch.float64) * 0.7).requires_grad_()
    m = torch.rand(1, 9, 3, 3, generator=g, dtype=torch.float64).requires_grad_()
    w = torch.randn(3, 2, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(3, generator=g, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda *a: bev.modulated_deform_conv(*a, stride=2, pad=1), (x, off, m, w, b),
                                    eps=1e-5, atol=1e-4, rtol=1e-3)


def _box(x, y, dx, dy, yaw, z=0.0, dz=1.0):
    return [x, y, z, dx, dy, dz, yaw]


def test_rotated_overlap_known_answers():
    a = torch.tensor([_box(0, 0, 4, 2, 0.0), _box(0, 0, 4, 2, 0.0), _box(0, 0, 4, 2, 0.0), _box(0, 0, 4, 2, 0.3),
                      _box(0, 0, 2, 2, 0.0), _box(1, 1, 2, 2, 0.0)])
    b = torch.tensor([_box(0, 0, 4, 2, 0.0), _box(10, 0, 4, 2, 0.0), _box(0, 0, 4, 2, math.pi / 2), _box(0, 0, 4, 2, 0.3),
                      _box(0, 0, 2, 2, math.pi / 4), _box(2, 2, 2, 2, 0.0)])
    ov = head.boxes_aligned_overlap_bev(a, b).numpy()
    oct_area = 8 * (math.sqrt(2) - 1)                                    # square /\ square rotated 45 deg = regular octagon
    np.testing.assert_allclose(ov, [8.0, 0.0, 4.0, 8.0, oct_area, 1.0], rtol=1e-4, atol=1e-4)
    iou = head.boxes_aligned_iou3d(a, b).view(-1).numpy()
    np.testing.assert_allclose(iou[:3], [1.0, 0.0, 4.0 / 12.0], rtol=1e-4, atol=1e-5)


def test_one_cycle_and_adam_step():
    lr0, m0 = optim.one_cycle(0, 1000)
    assert abs(lr0 - 1e-4) < 1e-12 and abs(m0 - 0.95) < 1e-12
    lr_pk, m_pk = optim.one_cycle(400, 1000)
    assert abs(lr_pk - 1e-3) < 1e-12 and abs(m_pk - 0.85) < 1e-12
    # vs torch.optim.Adam with the decoupled decay applied by hand (fastai_optim.py:135-152)
    g = torch.Generator().manual_seed(0)
    p = torch.randn(50, generator=g); grad = torch.randn(50, generator=g)
    q = torch.nn.Parameter(p.clone()); opt = torch.optim.Adam([q], lr=3e-4, betas=(0.93, 0.99))
    m = torch.zeros(50); v = torch.zeros(50); pp = p.clone()
    for step in (1, 2, 3):
        q.grad = grad.clone() * step
        q.data.mul_(1 - 0.01 * 3e-4); opt.step()
        optim.adam_true_wd_step([pp], [grad * step], [m], [v], step, 3e-4, 0.93)
    np.testing.assert_allclose(pp.numpy(), q.detach().numpy(), rtol=1e-6, atol=1e-7)


def test_nms_and_pairwise_iou_known_answers():
    """Greedy rotated NMS (iou3d_nms.cpp:137-183) and pairwise 3-D IoU (iou3d_nms_utils.py:55-81) on hand-checkable boxes."""
    from oracle import post
    b = torch.tensor([[0.0, 0, 0, 4, 2, 2, 0],      # kept
                      [1.0, 0, 0, 4, 2, 2, 0],      # overlap 3x2=6 with box 0: iou 6/(8+8-6)=0.6 -> suppressed at 0.5
                      [2.0, 0, 0, 4, 2, 2, 0],      # overlap with box 0: 2x2=4 -> iou 1/3 -> kept; it then suppresses nothing new
                      [20.0, 5, 0, 4, 2, 2, 0.3],   # far away, kept
                      [20.0, 5, 0, 4, 2, 2, 0.3]])  # duplicate of box 3 -> suppressed
    assert post.nms_bev_sorted(b, 0.5).tolist() == [0, 2, 3]
    assert post.nms_bev_sorted(b, 0.7).tolist() == [0, 1, 2, 3]            # 0.6 is below the threshold now
    assert post.nms_bev_sorted(b, 0.3).tolist() == [0, 3]                  # 1/3 > 0.3: box 2 goes too
    assert post.nms_bev_sorted(b[:0], 0.5).tolist() == []
    # box 1 is suppressed by box 0, so it must NOT suppress box 2 (iou(1,2) = 0.6): greedy, not transitive
    iou = post.boxes_iou3d(b[:3], b[:3])
    np.testing.assert_allclose(iou.numpy(), [[1, 0.6, 1 / 3], [0.6, 1, 0.6], [1 / 3, 0.6, 1]], rtol=1e-5)
    # height overlap: shift z by 1 of 2 -> 3-D overlap halves: 8/(16+16-8)
    c = b[:1].clone(); c[0, 2] = 1.0
    np.testing.assert_allclose(post.boxes_iou3d(b[:1], c).numpy(), [[8.0 / 24.0]], rtol=1e-5)
    scores = torch.tensor([0.9, 0.8, 0.7, 0.95, 0.1])
    keep, _ = post.nms_gpu(b, scores, 0.5)
    assert keep.tolist() == [3, 0, 2]                                      # descending score order of the survivors
    rec = post.recall_record(b[[0, 3]], torch.cat([b[[1, 3]], torch.zeros(2, 7)]), [0.3, 0.5, 0.7])
    assert rec == {"gt": 2, "rcnn_0.3": 2, "rcnn_0.5": 2, "rcnn_0.7": 1}


def test_hard_voxelizer_known_answers():
    """spconv Point2VoxelCPU3d semantics: first-appearance voxel order, stream order inside a voxel, both capacities, range filter."""
    from oracle import voxel as ovox
    rng = [0.0, 0.0, -1.0, 4.0, 4.0, 1.0]
    vs = [1.0, 1.0, 2.0]
    pts = np.array([[2.5, 0.5, 0.0, 7.0],     # voxel A (x2,y0) created
                    [0.5, 0.5, 0.0, 1.0],     # voxel B (x0,y0)
                    [2.6, 0.4, 0.1, 8.0],     # A, slot 1
                    [9.0, 0.5, 0.0, 0.0],     # outside -> dropped
                    [2.7, 0.3, 0.2, 9.0],     # A full (max_points 2) -> dropped
                    [3.5, 3.5, 0.0, 2.0],     # voxel C
                    [1.5, 1.5, 0.0, 3.0],     # would be voxel D but max_voxels = 3 -> dropped
                    [0.6, 0.6, 0.0, 4.0],     # B, slot 1
                    [4.0, 0.5, 0.0, 5.0]],    # x == range max -> floor(4/1) = 4 == grid -> dropped
                   dtype=np.float32)
    v, c, n = ovox.points_to_voxels(pts, vs, rng, max_points=2, max_voxels=3)
    assert c.tolist() == [[0, 0, 2], [0, 0, 0], [0, 3, 3]] and n.tolist() == [2, 2, 1]
    assert v[0, :, 3].tolist() == [7.0, 8.0] and v[1, :, 3].tolist() == [1.0, 4.0] and v[2, :, 3].tolist() == [2.0, 0.0]
    v0, c0, n0 = ovox.points_to_voxels(pts[:0], vs, rng, 2, 3)
    assert v0.shape == (0, 2, 4) and c0.shape == (0, 3) and n0.shape == (0,)
