"""Diagnostic (GPU box): cProfile of the host side of the training step (bf16x3 default), sorted by own time and by cumulative time.

    python tools/diag/host_cprofile.py [--batch 8] > gpurun_out/host_cprofile.log
"""
import argparse
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench as B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    from radardistill_amd import kernels as K
    from radardistill_amd.pcdet.models import model_fn_decorator
    from radardistill_amd.synthetic import make_batch
    from radardistill_amd.train import build_optimizer, build_scheduler
    K.set_conv_math(os.environ.get("RD_MATH", "bf16x3"))
    model, cfg, geom = B.build(os.path.join(ROOT, "tools/cfgs/radar_distill/bench_512.yaml"), args.grid, device)
    model.train()
    opt = build_optimizer(model, cfg.OPTIMIZATION)
    sched, _ = build_scheduler(opt, 100, 1, -1, cfg.OPTIMIZATION)
    fn = model_fn_decorator()
    batches = [B.device_batch(make_batch(batch_size=args.batch, n_lidar=35000, n_radar=2000, n_boxes=30, grid=args.grid, seed=i), device)
               for i in range(2)]

    def step(it):
        sched.step(it)
        opt.zero_grad()
        loss, tb, _ = fn(model, dict(batches[it % 2]))
        loss.backward()
        opt.step()

    for it in range(4):
        step(it)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for it in range(4, 4 + args.steps):
        step(it)
    pr.disable()
    torch.cuda.synchronize()
    for key in ("tottime", "cumtime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(70)
        print(f"==== sorted by {key} ({args.steps} steps)")
        print(s.getvalue())
    for pat in ("__getattr__", "_chk", "is_contiguous", "_call_impl"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_callers(pat)
        print(f"==== callers of {pat}")
        print("\n".join(l[:200] for l in s.getvalue().splitlines()[:60]))


if __name__ == "__main__":
    main()
