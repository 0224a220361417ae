"""Synthetic nuScenes-shaped radar + LiDAR sweeps (SURVEY.md section 8(d) generator spec).

Produces the `batch_dict` contract of the reference's `collate_batch`
(pcdet/datasets/dataset_distill.py:220-325): `points (sum N, 1+5)`, `radar_points (sum M, 1+6)`,
`gt_boxes (B, K, 10)`, `batch_size`; column 0 of the point arrays is the batch index.
"""
import numpy as np

# per-class nuScenes mean sizes (dx, dy, dz); order = CLASS_NAMES of radar_distill_train.yaml:1-2
_CLASS_DIMS = np.array([
    [4.63, 1.97, 1.74], [6.93, 2.51, 2.84], [6.37, 2.85, 3.19], [10.5, 2.94, 3.47], [12.29, 2.90, 3.87],
    [0.50, 2.53, 0.98], [2.11, 0.77, 1.47], [1.70, 0.60, 1.28], [0.73, 0.67, 1.77], [0.41, 0.41, 1.07]],
    dtype=np.float32)


def bench_geometry(grid=512):
    """Pillar geometry of the BASELINE configs: 0.2 m pillars, +-R m, z in [-5, 3]."""
    R = 0.1 * grid
    pc_range = [-R, -R, -5.0, R, R, 3.0]
    voxel_size = [0.2, 0.2, 0.2]
    grid_size = np.array([grid, grid, 40], dtype=np.int64)
    return pc_range, voxel_size, grid_size


def make_batch(batch_size=1, n_lidar=35000, n_radar=2000, n_boxes=30, grid=512, seed=0):
    rng = np.random.default_rng(seed)
    R = 0.1 * grid
    pts, rpts, boxes = [], [], []
    for b in range(batch_size):
        # LiDAR: r = |N(0, 18 m)| rejected beyond 1.4 R, theta uniform, z ~ N(-1, 1) clipped
        r = np.abs(rng.normal(0, 18.0, size=int(n_lidar * 1.3) + 16))
        r = r[r <= 1.4 * R][:n_lidar]
        while r.shape[0] < n_lidar:
            extra = np.abs(rng.normal(0, 18.0, size=n_lidar))
            r = np.concatenate([r, extra[extra <= 1.4 * R]])[:n_lidar]
        th = rng.uniform(0, 2 * np.pi, size=n_lidar)
        z = np.clip(rng.normal(-1.0, 1.0, size=n_lidar), -5, 3)
        inten = rng.uniform(0, 255, size=n_lidar)
        dt = rng.integers(0, 10, size=n_lidar) * 0.05
        p = np.stack([np.full(n_lidar, b), r * np.cos(th), r * np.sin(th), z, inten, dt], axis=1)
        pts.append(p.astype(np.float32))
        # radar: 70 % from 40 gaussian clusters (sigma 1.5 m), 30 % uniform
        centres = rng.uniform(-R, R, size=(40, 2))
        n_cl = int(0.7 * n_radar)
        cid = rng.integers(0, 40, size=n_cl)
        xy = np.concatenate([centres[cid] + rng.normal(0, 1.5, size=(n_cl, 2)),
                             rng.uniform(-R, R, size=(n_radar - n_cl, 2))], axis=0)
        zr = rng.normal(0.5, 0.2, size=n_radar)
        rcs = rng.uniform(-10, 40, size=n_radar)
        v = rng.normal(0, 5.0, size=(n_radar, 2))
        q = np.concatenate([np.full((n_radar, 1), b), xy, zr[:, None], rcs[:, None], v], axis=1)
        q = q[rng.permutation(n_radar)]
        rpts.append(q.astype(np.float32))
        # gt boxes: centres from the radar cluster centres
        cls = rng.integers(1, 11, size=n_boxes)
        c = centres[rng.integers(0, 40, size=n_boxes)] + rng.normal(0, 0.5, size=(n_boxes, 2))
        c = np.clip(c, -R + 1.0, R - 1.0)
        dims = _CLASS_DIMS[cls - 1] * rng.uniform(0.9, 1.1, size=(n_boxes, 3))
        zc = rng.normal(-0.5, 0.3, size=n_boxes)
        yaw = rng.uniform(-np.pi, np.pi, size=n_boxes)
        vel = rng.normal(0, 3.0, size=(n_boxes, 2))
        g = np.concatenate([c, zc[:, None], dims, yaw[:, None], vel, cls[:, None]], axis=1)
        boxes.append(g.astype(np.float32))
    return {
        "points": np.concatenate(pts, axis=0),
        "radar_points": np.concatenate(rpts, axis=0),
        "gt_boxes": np.stack(boxes, axis=0),
        "batch_size": batch_size,
    }
