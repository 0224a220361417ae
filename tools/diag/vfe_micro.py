"""Isolated timing of the segmented pillar VFE kernels (vfe_seg.hip) at the step's sizes: LiDAR 8 x 35 k points / radar 8 x 2 k points on
the 512 x 512 grid.  Prints us / launch of the statistics pass (train-mode BatchNorm, radar) and of the max pass, with the algorithmic
bytes of SURVEY 8(d): N (1 + C) 4 + P 32 4 + P 12.

    python tools/diag/vfe_micro.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from radardistill_amd import kernels as K      # noqa: E402
from radardistill_amd.synthetic import bench_geometry, make_batch      # noqa: E402


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    B, grid = 8, 512
    pc, voxel, gs = bench_geometry(grid)
    batch = make_batch(batch_size=B, n_lidar=35000, n_radar=2000, n_boxes=30, grid=grid, seed=0)
    for name, key in (("lidar", "points"), ("radar", "radar_points")):
        pts = torch.from_numpy(batch[key]).float().to(dev).contiguous()
        nf = pts.shape[1] - 1
        rg, point_row = K.voxelize(pts, B, grid, grid, pc[0], pc[1], voxel[0], voxel[1])
        P = int(K.rankgrid_count_tensor(rg, B * grid * grid))
        coords = K.rankgrid_coords(rg, B, grid, grid, True, P)
        offsets, order = K.vfe_group(point_row, P)
        g = torch.Generator().manual_seed(0)
        w = (torch.randn(32, 9 + nf, generator=g) * 0.1).to(dev)
        geom = torch.tensor([voxel[0], voxel[1], voxel[2], voxel[0] / 2 + pc[0], voxel[1] / 2 + pc[1], voxel[2] / 2 + pc[2], pc[0], pc[1], pc[2]],
                            dtype=torch.float32, device=dev)
        scale, shift = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
        by = pts.numel() * 4 + P * 32 * 4 + P * 12
        t0 = timed(lambda: K.vfe_seg_stats(pts, order, offsets, coords, w, geom, P))
        t1 = timed(lambda: K.vfe_seg_max(pts, order, offsets, coords, w, geom, scale, shift, P, True))
        tg = timed(lambda: K.vfe_group(point_row, P))
        print(f"{name}: {pts.shape[0]} points, {P} pillars ({pts.shape[0] / P:.2f} points / pillar): stats pass {t0:.1f} us, max pass {t1:.1f} us "
              f"({by / 1e6:.1f} MB algorithmic -> {by / t1 / 1e6:.2f} TB/s), grouping {tg:.1f} us")


if __name__ == "__main__":
    main()
