"""Diagnostic (GPU box): host enqueue time of the training forward by top-level module of the detector's module_list (+ the loss), no
device sync inside the step, host-bound batch by default.  Says where the ~5 ms of forward host time outside the autograd Functions go.

    python tools/diag/forward_modules.py [batch=1] [steps=40]
"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench as B

ACC = collections.defaultdict(lambda: [0, 0.0])


def timed(name, fn):
    pc = time.perf_counter

    def w(*a, **k):
        t = pc()
        try:
            return fn(*a, **k)
        finally:
            e = ACC[name]
            e[0] += 1
            e[1] += pc() - t
    return w


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    device = torch.device("cuda", 0)
    from radardistill_amd import autograd as A, kernels as K
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

    for it in range(6):
        step(it)
    torch.cuda.synchronize()
    for m in model.module_list:
        m.forward = timed("module " + m.__class__.__name__, m.forward)
        if hasattr(m, "prepare"):
            m.prepare = timed("prepare " + m.__class__.__name__, m.prepare)
        if hasattr(m, "get_loss"):
            m.get_loss = timed("get_loss " + m.__class__.__name__, m.get_loss)
        if hasattr(m, "assign_targets"):
            m.assign_targets = timed("assign_targets " + m.__class__.__name__, m.assign_targets)
    model._geometry_prelude = timed("geometry prelude", model._geometry_prelude)
    model.get_training_distll_loss = timed("get_training_distll_loss (all losses)", model.get_training_distll_loss)
    model.forward = timed("PillarNet.forward (total)", model.forward)
    A.begin_step = timed("begin_step", A.begin_step)
    A.end_forward = timed("end_forward", A.end_forward)
    t0 = time.perf_counter()
    for it in range(6, 6 + n):
        step(it)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"B = {batch}: {n} steps, host loop {host / n * 1e3:.2f} ms/step")
    for k, (c, t) in sorted(ACC.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:58s} {c / n:6.1f} calls  {t / n * 1e3:8.3f} ms/step")


if __name__ == "__main__":
    main()
