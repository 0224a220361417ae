"""Diagnostic (GPU box): device time of the ATen (non-librdamd) kernels of one training step, attributed to the ATen operator and
the innermost radardistill_amd source line that called it (torch.profiler with stacks).  Finds the elementwise / copy / cat
launches that still sit on the main stream."""
import collections
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
    n = 2
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for it in range(3, 3 + n):
            step(it)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        dt = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
        self_dt = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
        if not ev.name.startswith("aten::") or self_dt <= 0:
            continue
        where = "?"
        for fr in ev.stack or []:
            if "radardistill_amd" in fr and "diag" not in fr:
                where = fr.split("radardistill_amd/")[-1][:70]
                break
        a = agg[(ev.name, where)]
        a[0] += 1
        a[1] += self_dt
    tot = sum(v[1] for v in agg.values())
    print(f"ATen self device time: {tot / n / 1e3:.2f} ms/step over {sum(v[0] for v in agg.values()) / n:.0f} ops/step")
    for (name, where), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"  {t / n:8.1f} us/step  {c / n:6.1f} x  {name:32s} {where}")


if __name__ == "__main__":
    main()
