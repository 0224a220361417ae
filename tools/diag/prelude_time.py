"""Diagnostic (GPU box): host time of the geometry prelude (detectors/pillarnet.py::_geometry_prelude) per step, with the device idle when
it starts (so the number is host work + the one device->host read over a handful of small kernels), the separate entry points
(RD_GEOM_COMPOSITE=0) against rd_geometry_begin / rd_geometry_finish (=1), alternating inside one process.

    python tools/diag/prelude_time.py [batch=8] [rounds=30]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench as B


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    device = torch.device("cuda", 0)
    from radardistill_amd import autograd as A
    from radardistill_amd.synthetic import make_batch
    model, cfg, geom = B.build(os.path.join(ROOT, "tools/cfgs/radar_distill/bench_512.yaml"), 512, device)
    model.train()
    batches = [B.device_batch(make_batch(batch_size=batch, n_lidar=35000, n_radar=2000, n_boxes=30, grid=512, seed=i), device) for i in range(2)]
    acc = {"0": [], "1": []}
    for r in range(rounds + 4):
        for mode in ("0", "1"):
            os.environ["RD_GEOM_COMPOSITE"] = mode
            bd = dict(batches[r % 2])
            A.begin_step(device)
            torch.cuda.synchronize()
            t = time.perf_counter()
            model._geometry_prelude(bd, device)
            dt = time.perf_counter() - t
            torch.cuda.synchronize()
            if r >= 4:
                acc[mode].append(dt * 1e3)
    for mode, name in (("0", "separate entry points"), ("1", "rd_geometry_begin / _finish")):
        v = sorted(acc[mode])
        print(f"{name:32s}: median {v[len(v) // 2]:.3f} ms, min {v[0]:.3f} ms, max {v[-1]:.3f} ms per step (B = {batch}, both branches)")


if __name__ == "__main__":
    main()
