"""Oracle: 2-D sparse convolution (spconv 2.x semantics) and the PillarRes18 SparseEnc backbone.
SURVEY 8(a) rows A4, A5.  Test infrastructure.

spconv is a third-party dependency that is NOT vendored in the reference tree and is not installed
here (docs/INSTALL.md:39 pins `spconv-cu113`, version unpinned).  This file restates its published
semantics as used by the reference call sites
  pcdet/models/backbones_3d/spconv_backbone_2d.py:9-28   (post_act_block: SubMConv2d / SparseConv2d)
  pcdet/models/backbones_3d/spconv_backbone_2d.py:41-77  (SparseBasicBlock)
  pcdet/models/backbones_3d/spconv_backbone_2d.py:208-324 (PillarRes18BackBone8x)
  pcdet/models/backbones_3d/spconv_backbone_2d_distillation.py:6-96 (Radar_PillarRes18BackBone8x)
PARITY UNPINNED against the real spconv binary (no reference tests, library absent).  Pinned here
by the equivalence "sparse conv on active sites == dense conv on the zero-filled map, sampled at
the output sites" (tests/test_oracle_kat.py), which is spconv's defining property.

Canonical ordering: spconv's output-row order for SparseConv2d is implementation-defined (GPU hash
table); the canonical form used for bit-exact index / rulebook parity is rows sorted by (b, y, x).
Rulebook = for each of the 9 taps k = ky*3+kx, the list of (in_row, out_row) pairs sorted by out_row.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _keys(idx, H, W):
    return (idx[:, 0].astype(np.int64) * H + idx[:, 1]) * W + idx[:, 2]


def subm_rulebook(indices, spatial_shape):
    """SubMConv2d(k=3): out sites == in sites; tap (ky,kx) of output (y,x) reads input (y+ky-1, x+kx-1).
    indices (A,3) int32 (b,y,x).  Returns nbr (A,9) int32, -1 where the neighbour is inactive."""
    H, W = int(spatial_shape[0]), int(spatial_shape[1])
    idx = np.asarray(indices, dtype=np.int64)
    A = idx.shape[0]
    keys = _keys(idx, H, W)
    order = np.argsort(keys, kind="stable")
    skeys = keys[order]
    nbr = -np.ones((A, 9), dtype=np.int32)
    for ky in range(3):
        for kx in range(3):
            y = idx[:, 1] + ky - 1
            x = idx[:, 2] + kx - 1
            ok = (y >= 0) & (y < H) & (x >= 0) & (x < W)
            q = (idx[:, 0] * H + y) * W + x
            pos = np.searchsorted(skeys, q)
            pos = np.clip(pos, 0, max(A - 1, 0))
            hit = ok & (A > 0) & (skeys[pos] == q)
            nbr[hit, ky * 3 + kx] = order[pos[hit]]
    return nbr


def strided_rulebook(indices, spatial_shape, stride=2, pad=1, k=3):
    """SparseConv2d(k3, s2, p1): out shape floor((H+2p-k)/s)+1; output o is active iff some tap
    (ky,kx) and active input i satisfy o*s - p + k = i.  Returns (out_indices sorted by (b,y,x),
    out_shape, nbr (A_out, 9) int32 of input rows)."""
    H, W = int(spatial_shape[0]), int(spatial_shape[1])
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    idx = np.asarray(indices, dtype=np.int64)
    cand = []
    for ky in range(k):
        for kx in range(k):
            ny = idx[:, 1] + pad - ky
            nx = idx[:, 2] + pad - kx
            ok = (ny % stride == 0) & (nx % stride == 0)
            oy, ox = ny // stride, nx // stride
            ok &= (oy >= 0) & (oy < Ho) & (ox >= 0) & (ox < Wo)
            cand.append(np.stack([idx[ok, 0], oy[ok], ox[ok]], axis=1))
    cand = np.concatenate(cand, axis=0) if cand else np.zeros((0, 3), np.int64)
    okeys = np.unique(_keys(cand, Ho, Wo))
    out_idx = np.stack([okeys // (Ho * Wo), (okeys // Wo) % Ho, okeys % Wo], axis=1)
    # neighbour table: out (oy,ox), tap (ky,kx) -> in (oy*s - p + ky, ox*s - p + kx)
    A_in = idx.shape[0]
    ikeys = _keys(idx, H, W)
    order = np.argsort(ikeys, kind="stable")
    sk = ikeys[order]
    nbr = -np.ones((out_idx.shape[0], k * k), dtype=np.int32)
    for ky in range(k):
        for kx in range(k):
            y = out_idx[:, 1] * stride - pad + ky
            x = out_idx[:, 2] * stride - pad + kx
            ok = (y >= 0) & (y < H) & (x >= 0) & (x < W)
            q = (out_idx[:, 0] * H + y) * W + x
            pos = np.clip(np.searchsorted(sk, q), 0, max(A_in - 1, 0))
            hit = ok & (A_in > 0) & (sk[pos] == q)
            nbr[hit, ky * k + kx] = order[pos[hit]]
    return out_idx.astype(np.int32), (Ho, Wo), nbr


def pairs_from_nbr(nbr):
    """Canonical rulebook: list over taps of (in_row, out_row) int32 arrays, sorted by out_row."""
    out = []
    for t in range(nbr.shape[1]):
        o = np.nonzero(nbr[:, t] >= 0)[0].astype(np.int32)
        out.append(np.stack([nbr[o, t], o], axis=1))
    return out


def sparse_conv(feats, nbr, weight, bias=None):
    """out[j] = sum_t in[nbr[j,t]] @ W[:, t].T (+ bias).  weight layout [Cout, 3, 3, Cin] (spconv 2.x,
    detector3d_template.py:414-429).  Differentiable torch ops."""
    Cout = weight.shape[0]
    w = weight.reshape(Cout, 9, -1)
    nbr_t = torch.as_tensor(nbr, dtype=torch.int64)
    out = torch.zeros((nbr.shape[0], Cout), dtype=feats.dtype)
    for t in range(9):
        o = torch.nonzero(nbr_t[:, t] >= 0).squeeze(1)
        if o.numel() == 0:
            continue
        contrib = feats[nbr_t[o, t]] @ w[:, t, :].t()
        out = out.index_add(0, o, contrib)
    if bias is not None:
        out = out + bias
    return out


def bn1d(x, state, prefix, training, eps=1e-3, momentum=0.01):
    """nn.BatchNorm1d(eps=1e-3, momentum=0.01) over sparse rows (spconv_backbone_2d.py:212)."""
    return F.batch_norm(x, state[prefix + "running_mean"], state[prefix + "running_var"],
                        state[prefix + "weight"], state[prefix + "bias"], training=training,
                        momentum=momentum, eps=eps)


def sparse_basic_block(x, nbr, state, prefix, training):
    """SparseBasicBlock.forward (spconv_backbone_2d.py:61-77): conv(bias)->BN->ReLU->conv(bias)->BN->+id->ReLU."""
    out = sparse_conv(x, nbr, state[prefix + "conv1.weight"], state[prefix + "conv1.bias"])
    out = F.relu(bn1d(out, state, prefix + "bn1.", training))
    out = sparse_conv(out, nbr, state[prefix + "conv2.weight"], state[prefix + "conv2.bias"])
    out = bn1d(out, state, prefix + "bn2.", training)
    return F.relu(out + x)


def to_dense(feats, indices, batch_size, shape):
    """SparseConvTensor.dense(): zero-filled (B, C, H, W) with rows scattered (spconv_backbone_2d.py:299)."""
    H, W = shape
    C = feats.shape[1]
    idx = torch.as_tensor(np.asarray(indices), dtype=torch.int64)
    flat = (idx[:, 0] * H + idx[:, 1]) * W + idx[:, 2]
    dense = torch.zeros((batch_size * H * W, C), dtype=feats.dtype).index_add(0, flat, feats)
    return dense.view(batch_size, H, W, C).permute(0, 3, 1, 2).contiguous()


def bn2d(x, state, prefix, training, eps, momentum):
    return F.batch_norm(x, state[prefix + "running_mean"], state[prefix + "running_var"],
                        state[prefix + "weight"], state[prefix + "bias"], training=training,
                        momentum=momentum, eps=eps)


def dense_basic_block(x, state, prefix, training):
    """BasicBlock.forward (spconv_backbone_2d.py:80-112)."""
    out = F.conv2d(x, state[prefix + "conv1.weight"], state[prefix + "conv1.bias"], padding=1)
    out = F.relu(bn2d(out, state, prefix + "bn1.", training, 1e-3, 0.01))
    out = F.conv2d(out, state[prefix + "conv2.weight"], state[prefix + "conv2.bias"], padding=1)
    out = bn2d(out, state, prefix + "bn2.", training, 1e-3, 0.01)
    return F.relu(out + x)


def pillar_res18_backbone(pillar_features, pillar_coords, batch_size, grid_size, state, prefix, training):
    """PillarRes18BackBone8x.forward (spconv_backbone_2d.py:261-324) == Radar_PillarRes18BackBone8x.forward
    (spconv_backbone_2d_distillation.py:59-96).  sparse_shape = grid_size[[1, 0]] (:213).

    Returns dict with x_conv1..3 as (features, indices, shape), x_conv4 dense (B,256,H/8,W/8),
    x_conv5 (B,256,H/16,W/16), and `rulebooks` (nbr tables per stage) for index parity."""
    shape = (int(grid_size[1]), int(grid_size[0]))
    idx = np.asarray(pillar_coords, dtype=np.int32)
    x = pillar_features
    out, books = {}, {}
    nbr = subm_rulebook(idx, shape)
    books["res1"] = nbr
    for b in range(2):
        x = sparse_basic_block(x, nbr, state, f"{prefix}conv1.{b}.", training)
    out["x_conv1"] = (x, idx, shape)
    for stage in (2, 3, 4):
        oidx, oshape, snbr = strided_rulebook(idx, shape)
        books[f"spconv{stage}"] = snbr
        x = sparse_conv(x, snbr, state[f"{prefix}conv{stage}.0.0.weight"])
        x = F.relu(bn1d(x, state, f"{prefix}conv{stage}.0.1.", training))
        idx, shape = oidx, oshape
        nbr = subm_rulebook(idx, shape)
        books[f"res{stage}"] = nbr
        for b in (1, 2):
            x = sparse_basic_block(x, nbr, state, f"{prefix}conv{stage}.{b}.", training)
        out[f"x_conv{stage}"] = (x, idx, shape)
    x4 = to_dense(x, idx, batch_size, shape)
    out["x_conv4_sparse"] = out["x_conv4"]
    out["x_conv4"] = x4
    y = F.conv2d(x4, state[prefix + "conv5.0.0.weight"], None, stride=2, padding=1)
    y = F.relu(bn2d(y, state, prefix + "conv5.0.1.", training, 1e-3, 0.01))
    y = dense_basic_block(y, state, prefix + "conv5.1.", training)
    y = dense_basic_block(y, state, prefix + "conv5.2.", training)
    out["x_conv5"] = y
    out["rulebooks"] = books
    return out
