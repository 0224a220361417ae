"""Oracle: dynamic pillar voxelization + PillarVFE.  Test infrastructure.

Follows pcdet/models/backbones_3d/vfe/dynamic_pillar_vfe.py:195-252 (LiDAR) / :256-313 (radar;
identical arithmetic, different batch_dict keys) and PFNLayerV2 :14-46.
"""
import torch
import torch.nn.functional as F


def voxelize(points, pc_range, voxel_size, grid_xy):
    """dynamic_pillar_vfe.py:200-212.  points (N, 1+C) fp32, column 0 = batch index.

    Returns (kept_points, points_coords (n,2) int32 [cx, cy], unq_keys (P,) sorted int32,
    unq_inv (n,) int64, unq_cnt (P,)).
    The division is a true fp32 division (not a reciprocal multiply), as in the reference (:201-202).
    """
    rng = torch.as_tensor(pc_range, dtype=torch.float32)
    vs = torch.as_tensor(voxel_size, dtype=torch.float32)
    grid = torch.as_tensor(list(grid_xy), dtype=torch.int64)
    coords = torch.floor((points[:, [1, 2]] - rng[[0, 1]]) / vs[[0, 1]]).int()
    mask = ((coords >= 0) & (coords < grid)).all(dim=1)
    points = points[mask]
    coords = coords[mask]
    scale_xy = int(grid_xy[0]) * int(grid_xy[1])
    scale_y = int(grid_xy[1])
    merge = points[:, 0].int() * scale_xy + coords[:, 0] * scale_y + coords[:, 1]
    unq, inv, cnt = torch.unique(merge, return_inverse=True, return_counts=True, dim=0)
    return points, coords, unq.int(), inv, cnt


def segment_mean(src, index, n):
    """torch_scatter.scatter_mean semantics (dynamic_pillar_vfe.py:226)."""
    out = torch.zeros((n, src.shape[1]), dtype=src.dtype)
    out.index_add_(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones(len(index), dtype=src.dtype))
    return out / cnt.clamp(min=1).unsqueeze(-1)


def segment_max(src, index, n):
    """torch_scatter.scatter_max(...)[0] semantics (dynamic_pillar_vfe.py:40); differentiable
    (gradient goes to an arg-max element)."""
    out = torch.full((n, src.shape[1]), float("-inf"), dtype=src.dtype)
    return out.scatter_reduce(0, index.unsqueeze(-1).expand_as(src), src, reduce="amax", include_self=True)


def point_features(points, coords, inv, n_pillars, pc_range, voxel_size):
    """dynamic_pillar_vfe.py:214-237 with USE_ABSLOTE_XYZ, USE_CLUSTER_XYZ, USE_RELATIVE_XYZ True,
    WITH_DISTANCE False (radar_distill_train.yaml:70-84).  Concat order:
    [f_center(3), raw point features incl. xyz (C), f_cluster(3), f_relative(3)]."""
    vx, vy, vz = [float(v) for v in voxel_size]
    x_off = vx / 2 + float(pc_range[0])
    y_off = vy / 2 + float(pc_range[1])
    z_off = vz / 2 + float(pc_range[2])
    xyz = points[:, [1, 2, 3]].contiguous()
    f_center = torch.zeros_like(xyz)
    f_center[:, 0] = xyz[:, 0] - (coords[:, 0].to(xyz.dtype) * vx + x_off)
    f_center[:, 1] = xyz[:, 1] - (coords[:, 1].to(xyz.dtype) * vy + y_off)
    f_center[:, 2] = xyz[:, 2] - z_off
    mean = segment_mean(xyz, inv, n_pillars)
    f_cluster = xyz - mean[inv]
    f_rel = xyz - torch.as_tensor(pc_range[:3], dtype=torch.float32)
    return torch.cat([f_center, points[:, 1:], f_cluster, f_rel], dim=-1)


def pfn_layer(feats, inv, n_pillars, state, prefix, training):
    """PFNLayerV2 as the single / last layer (dynamic_pillar_vfe.py:14-46): Linear(no bias) ->
    BatchNorm1d(eps 1e-3, momentum 0.01) -> ReLU -> segment max."""
    x = F.linear(feats, state[prefix + "linear.weight"])
    x = F.batch_norm(x, state[prefix + "norm.running_mean"], state[prefix + "norm.running_var"],
                     state[prefix + "norm.weight"], state[prefix + "norm.bias"],
                     training=training, momentum=0.01, eps=1e-3)
    x = F.relu(x)
    return segment_max(x, inv, n_pillars)


def dynamic_pillar_vfe(points, state, prefix, pc_range, voxel_size, grid_size, training):
    """Full forward of (Radar_)DynamicPillarVFESimple2D.

    Returns dict(pillar_features (P,32), pillar_coords (P,3) int32 (b, y, x), unq_inv (n,) int64,
    point_mask-kept points)."""
    gx, gy = int(grid_size[0]), int(grid_size[1])
    pts, coords, unq, inv, cnt = voxelize(points, pc_range, voxel_size, (gx, gy))
    P = int(unq.shape[0])
    feats = point_features(pts, coords, inv, P, pc_range, voxel_size)
    out = pfn_layer(feats, inv, P, state, prefix + "pfn_layers.0.", training)
    scale_xy, scale_y = gx * gy, gy
    # :243-248  decode key -> (b, cx, cy) then reorder [0, 2, 1] -> (b, y, x)
    pc = torch.stack((unq // scale_xy, (unq % scale_xy) // scale_y, unq % scale_y), dim=1)
    pc = pc[:, [0, 2, 1]].int()
    return {"pillar_features": out, "pillar_coords": pc, "unq_inv": inv, "unq_cnt": cnt,
            "point_features": feats, "kept_points": pts}
