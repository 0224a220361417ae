"""Diagnostic (GPU box): which Python / autograd call paths launch the remaining ATen copy / add / cat kernels of a training step
(torch.profiler, grouped by the top stack frames)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B                                                              # noqa: E402


def main():
    device = torch.device("cuda", 0)
    from radardistill_amd import kernels as K
    from radardistill_amd.pcdet.models import model_fn_decorator
    from radardistill_amd.synthetic import make_batch
    from radardistill_amd.train import build_optimizer, build_scheduler
    K.set_conv_math("bf16x3")
    model, cfg, geom = B.build(os.path.join(ROOT, "tools/cfgs/radar_distill/bench_512.yaml"), 512, device)
    model.train()
    opt = build_optimizer(model, cfg.OPTIMIZATION)
    sched, _ = build_scheduler(opt, 100, 1, -1, cfg.OPTIMIZATION)
    fn = model_fn_decorator()
    batches = [B.device_batch(make_batch(batch_size=8, n_lidar=35000, n_radar=2000, n_boxes=30, grid=512, seed=i), device) for i in range(2)]

    def step(it):
        sched.step(it)
        opt.zero_grad()
        loss, tb, _ = fn(model, dict(batches[it % 2]))
        loss.backward()
        opt.step()

    for it in range(3):
        step(it)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step(3)
        torch.cuda.synchronize()
    want = ("aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::sum", "aten::mul", "aten::fill_", "aten::contiguous", "aten::clone")
    rows = []
    for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
        dt = getattr(ev, "self_device_time_total", 0)
        if ev.key in want and dt > 0:
            rows.append((dt, ev.count, ev.key, str(ev.input_shapes)[:60], [s for s in ev.stack][:6]))
    rows.sort(reverse=True)
    for dt, n, key, shp, stack in rows[:40]:
        print(f"{dt:9.1f} us  {n:3d} x  {key:16s} {shp}")
        for s in stack:
            if "profiler" not in s:
                print("             ", s[-110:])


if __name__ == "__main__":
    main()
