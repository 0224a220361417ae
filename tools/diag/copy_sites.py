"""Diagnostic (GPU box): which Python call sites make `.contiguous()` COPY (a launch + the bytes) in a training step, and how much.

    python tools/diag/copy_sites.py [batch=8]
"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench as B


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    device = torch.device("cuda", 0)
    from radardistill_amd import kernels as K
    from radardistill_amd.pcdet.models import model_fn_decorator
    from radardistill_amd.synthetic import make_batch
    from radardistill_amd.train import build_optimizer, build_scheduler
    K.set_conv_math("bf16x3")
    model, cfg, geom = B.build(os.path.join(ROOT, "tools/cfgs/radar_distill/bench_512.yaml"), 512, device)
    model.train()
    opt = build_optimizer(model, cfg.OPTIMIZATION)
    sched, _ = build_scheduler(opt, 1000, 1, -1, cfg.OPTIMIZATION)
    fn = model_fn_decorator()
    batches = [B.device_batch(make_batch(batch_size=batch, n_lidar=35000, n_radar=2000, n_boxes=30, grid=512, seed=i), device) for i in range(2)]

    def step(it):
        sched.step(it)
        opt.zero_grad()
        loss, tb, _ = fn(model, dict(batches[it % 2]))
        loss.backward()
        opt.step()

    for it in range(3):
        step(it)
    torch.cuda.synchronize()
    sites = collections.defaultdict(lambda: [0, 0])
    orig = torch.Tensor.contiguous

    def contiguous(self, *a, **k):
        if not self.is_contiguous():
            st = traceback.extract_stack(limit=4)[:-1]
            key = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(st))
            v = sites[key]
            v[0] += 1
            v[1] += self.numel() * self.element_size()
        return orig(self, *a, **k)

    torch.Tensor.contiguous = contiguous
    n = 4
    try:
        for it in range(3, 3 + n):
            step(it)
        torch.cuda.synchronize()
    finally:
        torch.Tensor.contiguous = orig
    for key, (c, by) in sorted(sites.items(), key=lambda kv: -kv[1][1]):
        print(f"{c / n:6.1f} copies/step  {by / n / 1e6:9.2f} MB/step  {key}")
    if not sites:
        print("no copying .contiguous() call")


if __name__ == "__main__":
    main()
