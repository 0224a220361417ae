"""Timings of the SURVEY 8(f) rows built after the training hot path (not the BASELINE metric; bench.py stays the contract):
eval forward (student + teacher chain, decode, rotated NMS, recall) and the padded-voxel front end (hard voxeliser + PillarVFE +
scatter).  Prints one JSON line per row.

    python tools/bench_next.py [--batch 8] [--grid 512]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench as B


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--grid", type=int, default=512)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    from radardistill_amd import kernels as K
    from radardistill_amd.pcdet.config import AttrDict
    from radardistill_amd.pcdet.models.backbones_2d.map_to_bev import __all__ as M2B
    from radardistill_amd.pcdet.models.backbones_3d.vfe import __all__ as VFE
    from radardistill_amd.pcdet.ops.iou3d_nms import iou3d_nms_utils as U
    from radardistill_amd.synthetic import bench_geometry, make_batch
    from radardistill_amd.voxel import VoxelGenerator
    K.set_conv_math("bf16x3")
    model, cfg, geom = B.build(os.path.join(ROOT, "tools/cfgs/radar_distill/bench_512.yaml"), args.grid, dev)
    model.eval()
    batch = B.device_batch(make_batch(batch_size=args.batch, n_lidar=35000, n_radar=2000, n_boxes=30, grid=args.grid, seed=0), dev)

    def eval_step():
        with torch.no_grad():
            return model(dict(batch))

    dt = timed(eval_step, n=10)
    preds, recall = eval_step()
    print(json.dumps({"row": "8(f) rank 2: eval forward + decode + rotated NMS + recall", "samples_per_sec": round(args.batch / dt, 2),
                      "ms_per_batch": round(dt * 1e3, 2), "batch": args.batch, "detections_sample0": int(preds[0]["pred_boxes"].shape[0]), "conv_math": "bf16x3"}))
    g = np.random.default_rng(0)
    n = 1000
    centres = g.uniform(-50, 50, size=(n // 5, 2))
    c = centres[g.integers(0, n // 5, size=n)] + g.normal(0, 1.0, size=(n, 2))
    boxes = torch.from_numpy(np.concatenate([c, np.zeros((n, 1)), g.uniform(2, 5, (n, 1)), g.uniform(1, 2.5, (n, 1)), np.full((n, 1), 1.5),
                                             g.uniform(-3.14, 3.14, (n, 1))], 1).astype(np.float32)).to(dev)
    scores = torch.from_numpy(g.uniform(0, 1, n).astype(np.float32)).to(dev)
    order = scores.sort(descending=True)[1]
    bs = boxes[order].contiguous()
    dt = timed(lambda: K.nms_bev(bs, 0.2), n=50)
    keep, num = K.nms_bev(bs, 0.2)
    print(json.dumps({"row": "rd_nms_bev (bit matrix + greedy pass on the device)", "boxes": n, "kept": int(num.item()), "us_per_call": round(dt * 1e6, 1)}))

    pc_range, voxel, grid = bench_geometry(args.grid)
    voxel = [voxel[0], voxel[1], pc_range[5] - pc_range[2]]
    pts = torch.from_numpy(make_batch(batch_size=args.batch, n_lidar=35000, n_radar=16, n_boxes=2, grid=args.grid, seed=1)["points"]).to(dev)
    gen = VoxelGenerator(voxel, pc_range, 5, 20, 30000)          # nuScenes pillar configs: 20 points, 30000 voxels
    dt_v = timed(lambda: gen.generate(pts, batch_size=args.batch), n=10)
    v, cds, num = gen.generate(pts, batch_size=args.batch)
    vfe = VFE["PillarVFE"](AttrDict(USE_NORM=True, WITH_DISTANCE=False, USE_ABSLOTE_XYZ=True, NUM_FILTERS=[64]), num_point_features=5,
                           voxel_size=voxel, point_cloud_range=pc_range).to(dev).eval()
    sc = M2B["PointPillarScatter"](AttrDict(NUM_BEV_FEATURES=64), grid_size=[int(grid[0]), int(grid[1]), 1])

    def front():
        with torch.no_grad():
            return sc(vfe({"voxels": v, "voxel_num_points": num, "voxel_coords": cds, "batch_size": args.batch}))

    dt_f = timed(front, n=20)
    # the CPU algorithm the voxeliser replaces, on one sample
    from oracle import voxel as ovox
    p0 = pts[pts[:, 0] == 0][:, 1:].cpu().numpy()
    t0 = time.perf_counter(); ovox.points_to_voxels(p0, voxel, pc_range, 20, 30000); dt_cpu = time.perf_counter() - t0
    print(json.dumps({"row": "8(f) rank 3: hard voxeliser (8 x 35k points, 20 pts/voxel, 30000 voxels/sample)", "ms_per_batch": round(dt_v * 1e3, 3),
                      "points_per_sec": round(pts.shape[0] / dt_v), "voxels": int(v.shape[0]),
                      "cpu_oracle_ms_per_sample": round(dt_cpu * 1e3, 1), "note": "the oracle is a python loop (spconv's C++ loop is absent here)"}))
    print(json.dumps({"row": "8(f) rank 3: PillarVFE + PointPillarScatter", "ms_per_batch": round(dt_f * 1e3, 3), "voxels": int(v.shape[0])}))


if __name__ == "__main__":
    main()
