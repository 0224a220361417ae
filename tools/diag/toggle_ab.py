"""Diagnostic (GPU box): A/B of a run-time switch INSIDE one process, alternating step by step, so that the box's CPU noise (the same
build's host loop was seen between 13.2 and 19.3 ms per step on one box within a minute) hits both arms alike.  Reports the median
and the minimum host time of a step per arm.

    python tools/diag/toggle_ab.py composite [batch=1] [steps=200]          (autograd.COMPOSITE: one library call per layer and direction)
    python tools/diag/toggle_ab.py nochk                                     (kernels._chk replaced by a no-op: what argument validation costs)
"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench as B


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "composite"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    device = torch.device("cuda", 0)
    from radardistill_amd import autograd as A, kernels as K
    from radardistill_amd.pcdet.models import model_fn_decorator
    from radardistill_amd.synthetic import make_batch
    from radardistill_amd.train import build_optimizer, build_scheduler
    K.set_conv_math("bf16x3")
    model, cfg, geom = B.build(os.path.join(ROOT, "tools/cfgs/radar_distill/bench_512.yaml"), 512, device)
    model.train()
    opt = build_optimizer(model, cfg.OPTIMIZATION)
    sched, _ = build_scheduler(opt, 10000, 1, -1, cfg.OPTIMIZATION)
    fn = model_fn_decorator()
    batches = [B.device_batch(make_batch(batch_size=batch, n_lidar=35000, n_radar=2000, n_boxes=30, grid=512, seed=i), device) for i in range(2)]

    _chk0 = K._chk

    def _nochk(t, *a, **k):
        return t

    def set_arm(arm):
        if what == "composite":
            A.COMPOSITE[0] = bool(arm)
        elif what == "rows":          # autograd.ROWS_SHORTCUT: consecutive layers pass the rows tensor along
            A.ROWS_SHORTCUT[0] = bool(arm)
        elif what == "hot":          # autograd.HOT_CACHES: cached geometry specs and BatchNorm tensor lookups
            A.HOT_CACHES[0] = bool(arm)
        elif what == "attrs":          # autograd.fast_module_attrs: submodules / parameters mirrored into the instance dictionaries
            A.fast_module_attrs(model, on=bool(arm))
        elif what == "nochk":          # how much the wrappers' argument validation costs (kernels._chk): an upper bound for caching it
            K._chk = _nochk if arm else _chk0
        else:
            raise SystemExit(f"unknown switch {what}")

    times = {0: [], 1: []}
    for it in range(n + 8):
        arm = it & 1
        set_arm(arm)
        torch.cuda.synchronize()          # every step starts with the device idle: the time below is host work only
        t0 = time.perf_counter()
        sched.step(it)
        opt.zero_grad()
        loss, tb, _ = fn(model, dict(batches[(it >> 1) % 2]))
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        if it >= 8:
            times[arm].append(dt * 1e3)
    for arm in (0, 1):
        v = times[arm]
        print(f"{what} = {arm}: host time per step median {statistics.median(v):.3f} ms, min {min(v):.3f} ms, mean {statistics.fmean(v):.3f} ms  (B = {batch}, {len(v)} steps)")


if __name__ == "__main__":
    main()
