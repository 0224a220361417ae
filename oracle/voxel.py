"""CPU oracle of the padded-voxel input format.  TEST INFRASTRUCTURE, NOT PRODUCT.

  * points_to_voxels: spconv's Point2VoxelCPU3d as called by VoxelGeneratorWrapper (pcdet/datasets/processor/data_processor.py:16-61).
    spconv (third-party, pinned by the reference's docs/INSTALL.md to spconv 2.x) is absent here, so this follows its published
    algorithm (spconv/csrc/sparse/pointops.py `points_to_voxel_3d_np`: per point c = floor((p - range_min) / vsize), zyx order;
    new voxel unless max_voxels reached; append while fewer than max_points) -- "parity unpinned" vs the binary.
  * pillar_vfe / scatter: pcdet/models/backbones_3d/vfe/pillar_vfe.py:8-123, pointpillar_scatter.py:5-37, pinned by
    tests/golden/g7_pillar.npz generated from those leaf modules.
"""
import numpy as np
import torch
import torch.nn.functional as F


def points_to_voxels(points, vsize_xyz, range_xyz, max_points, max_voxels):
    """points (n, C) float32 of ONE sample -> voxels (M, max_points, C), coords (M, 3) int32 (z, y, x), num_points (M,) int32."""
    points = np.asarray(points, dtype=np.float32)
    vs = np.asarray(vsize_xyz, dtype=np.float32)
    lo = np.asarray(range_xyz[:3], dtype=np.float32)
    grid = np.round((np.asarray(range_xyz[3:6], dtype=np.float64) - np.asarray(range_xyz[:3], dtype=np.float64)) / np.asarray(vsize_xyz, dtype=np.float64)).astype(np.int64)
    C = points.shape[1]
    voxels = np.zeros((max_voxels, max_points, C), dtype=np.float32)
    coords = np.zeros((max_voxels, 3), dtype=np.int32)
    num = np.zeros(max_voxels, dtype=np.int32)
    index = {}
    m = 0
    for i in range(points.shape[0]):
        c = np.floor((points[i, :3] - lo) / vs).astype(np.int64)          # fp32 arithmetic, as in the C++ loop
        if (c < 0).any() or (c >= grid).any():
            continue
        key = (int(c[2]), int(c[1]), int(c[0]))
        v = index.get(key, -1)
        if v == -1:
            if m >= max_voxels:
                continue
            v = m
            m += 1
            index[key] = v
            coords[v] = key
        if num[v] < max_points:
            voxels[v, num[v]] = points[i]
            num[v] += 1
    return voxels[:m], coords[:m], num[:m]


def batch_points_to_voxels(points_b, batch, vsize_xyz, range_xyz, max_points, max_voxels):
    """points_b (N, 1+C) with the batch id first -> concatenated (voxels, coords (M,4) = (b,z,y,x), num_points), as collate_batch does."""
    vs, cs, ns = [], [], []
    for b in range(batch):
        v, c, n = points_to_voxels(points_b[points_b[:, 0] == b][:, 1:], vsize_xyz, range_xyz, max_points, max_voxels)
        vs.append(v); ns.append(n)
        cs.append(np.concatenate([np.full((c.shape[0], 1), b, dtype=np.int32), c], axis=1))
    return np.concatenate(vs), np.concatenate(cs), np.concatenate(ns)


def pillar_vfe(voxels, num_points, coords, state, voxel_size, pc_range, use_abs_xyz=True, with_distance=False, training=False, prefix="",
               new_running=None):
    """PillarVFE.forward (pillar_vfe.py:85-123) with its stack of PFNLayers (pillar_vfe.py:29-49: Linear [-> BatchNorm1d over all
    M*P slots] -> ReLU -> max over the slots; a non-last layer returns [x | max repeated]) -> (M, Cout).  The layer count and
    USE_NORM are read off the state names; `new_running` (dict) receives the updated running statistics in training mode."""
    vx, vy, vz = voxel_size
    xo, yo, zo = vx / 2 + pc_range[0], vy / 2 + pc_range[1], vz / 2 + pc_range[2]
    mean = voxels[:, :, :3].sum(dim=1, keepdim=True) / num_points.type_as(voxels).view(-1, 1, 1)
    f_cluster = voxels[:, :, :3] - mean
    f_center = torch.zeros_like(voxels[:, :, :3])
    f_center[:, :, 0] = voxels[:, :, 0] - (coords[:, 3].to(voxels.dtype).unsqueeze(1) * vx + xo)
    f_center[:, :, 1] = voxels[:, :, 1] - (coords[:, 2].to(voxels.dtype).unsqueeze(1) * vy + yo)
    f_center[:, :, 2] = voxels[:, :, 2] - (coords[:, 1].to(voxels.dtype).unsqueeze(1) * vz + zo)
    feats = [voxels if use_abs_xyz else voxels[..., 3:], f_cluster, f_center]
    if with_distance:
        feats.append(torch.norm(voxels[:, :, :3], 2, 2, keepdim=True))
    feats = torch.cat(feats, dim=-1)
    P = feats.shape[1]
    mask = (num_points.int().unsqueeze(1) > torch.arange(P, dtype=torch.int).view(1, -1)).unsqueeze(-1).type_as(voxels)
    x = feats * mask
    n_layers = 0
    while f"{prefix}pfn_layers.{n_layers}.linear.weight" in state:
        n_layers += 1
    for i in range(n_layers):
        q = f"{prefix}pfn_layers.{i}."
        x = F.linear(x, state[q + "linear.weight"], state.get(q + "linear.bias"))
        if q + "norm.weight" in state:
            rm, rv = state[q + "norm.running_mean"].clone(), state[q + "norm.running_var"].clone()
            x = F.batch_norm(x.permute(0, 2, 1), rm, rv, state[q + "norm.weight"], state[q + "norm.bias"], training, 0.01, 1e-3).permute(0, 2, 1)
            if new_running is not None:
                new_running[q + "norm.running_mean"], new_running[q + "norm.running_var"] = rm, rv
        x = F.relu(x)
        x_max = torch.max(x, dim=1, keepdim=True)[0]
        x = x_max if i == n_layers - 1 else torch.cat([x, x_max.repeat(1, P, 1)], dim=2)
    return x.squeeze(1)


def scatter(pillar_features, coords, batch, nx, ny):
    """PointPillarScatter.forward (pointpillar_scatter.py:14-37) -> (B, C, ny, nx)."""
    C = pillar_features.shape[1]
    out = torch.zeros((batch, C, ny * nx), dtype=pillar_features.dtype)
    for b in range(batch):
        m = coords[:, 0] == b
        idx = (coords[m, 1] + coords[m, 2] * nx + coords[m, 3]).long()
        out[b][:, idx] = pillar_features[m].t()
    return out.view(batch, C, ny, nx)
