"""Diagnostic (GPU box): where does the HOST time of one eager training step go?

Runs the bench workload, then profiles 2 steps with torch.profiler (CPU + device activities, Python stacks) and prints
  * ops sorted by call count / self CPU time,
  * for the ATen ops that launch small kernels (copy_, add, fill_, mul, sum ...) the Python call sites, grouped by stack.
Also times a step with the device idle at the start (sync before the step): the host-only enqueue time.

    python tools/diag/host_profile.py [--batch 8] [--grid 512] > gpurun_out/host_profile.log
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench as B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--stacks", type=int, default=1)
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    from radardistill_amd import kernels as K
    K.set_conv_math(os.environ.get("RD_MATH", "bf16x3"))
    from radardistill_amd.pcdet.models import model_fn_decorator
    from radardistill_amd.synthetic import make_batch
    from radardistill_amd.train import build_optimizer, build_scheduler
    model, cfg, geom = B.build(os.path.join(ROOT, "tools/cfgs/radar_distill/bench_512.yaml"), args.grid, device)
    model.train()
    opt = build_optimizer(model, cfg.OPTIMIZATION)
    sched, _ = build_scheduler(opt, 100, 1, -1, cfg.OPTIMIZATION)
    fn = model_fn_decorator()
    if os.environ.get("RD_DP_REHEARSE") == "1":          # the data-parallel path in a world of one rank (dist.rehearsal)
        from radardistill_amd import dist as D
        D.init_distributed(backend="nccl", device=device)
        model = D.data_parallel(model, opt, 0)
    batches = [B.device_batch(make_batch(batch_size=args.batch, n_lidar=35000, n_radar=2000, n_boxes=30, grid=args.grid, seed=i), device)
               for i in range(2)]

    def step(it, timing=None):
        t0 = time.perf_counter()
        sched.step(it)
        opt.zero_grad()
        loss, tb, _ = fn(model, dict(batches[it % 2]))
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        opt.step()
        t3 = time.perf_counter()
        if timing is not None:
            timing.append((t1 - t0, t2 - t1, t3 - t2))
        return loss

    for it in range(4):
        step(it)
    torch.cuda.synchronize()
    # host-only enqueue time: device idle at the start of every step, so the in-step syncs wait for little
    tm = []
    for it in range(4, 8):
        torch.cuda.synchronize()
        step(it, tm)
    torch.cuda.synchronize()
    for a, b, c in tm:
        print(f"[host, device idle at step start] fwd+loss {a * 1e3:.1f} ms  bwd {b * 1e3:.1f} ms  opt {c * 1e3:.1f} ms", flush=True)

    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=bool(args.stacks), record_shapes=False) as prof:
        for it in range(8, 10):
            step(it)
        torch.cuda.synchronize()
    ka = prof.key_averages()
    print("==== by self CPU time")
    print(ka.table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=70))
    print("==== by count")
    rows = sorted(ka, key=lambda e: -e.count)[:60]
    for e in rows:
        print(f"{e.count / 2:8.1f}/step  self_cpu {e.self_cpu_time_total / 2e3:8.2f} ms/step  {e.key[:100]}")
    if args.stacks:
        kas = prof.key_averages(group_by_stack_n=8)
        want = ("aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::mul", "aten::sum", "aten::contiguous", "aten::clone",
                "aten::zeros", "aten::empty", "aten::to", "aten::_to_copy", "aten::item", "aten::_local_scalar_dense", "aten::cat", "aten::div",
                "aten::neg", "aten::index", "aten::where", "aten::pad", "aten::constant_pad_nd", "aten::zeros_like", "aten::empty_like")
        print("==== call sites of small ATen ops (count/step, op, innermost repo frames)")
        sites = []
        for e in kas:
            if e.key in want and e.count >= 2:
                st = [s for s in e.stack if "/repo/" in s or "radardistill_amd" in s or "bench.py" in s][:4]
                sites.append((e.count / 2, e.key, " <- ".join(s.split("/repo/")[-1] for s in st)))
        for c, k, s in sorted(sites, key=lambda t: -t[0])[:120]:
            print(f"{c:7.1f}  {k:28s} {s}")


if __name__ == "__main__":
    main()
