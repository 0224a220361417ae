"""Diagnostic (GPU box): host time of the training step by layer of the software stack, measured with perf_counter wrappers that work in
every thread (the autograd engine runs the backward functions in its own thread, where cProfile does not look):

  * each C-ABI entry point (`native.lib().rd_*`): the time inside the library = argument conversion + hipLaunchKernel,
  * each wrapper of radardistill_amd/kernels.py (checks, output allocation, the C call),
  * forward / backward of every torch.autograd.Function of radardistill_amd/autograd.py,
  * torch.empty / empty_like / zeros (allocator).

Nested wrappers are inclusive (a kernels.py wrapper contains its C call).  Run at a small batch so that the step is host-bound:

    python tools/diag/host_timers.py [batch=1] [steps=40] > gpurun_out/host_timers.log
"""
import collections
import inspect
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
    w.__wrapped__ = fn
    return w


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    device = torch.device("cuda", 0)
    from radardistill_amd import autograd as A, kernels as K, native
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
    phase = collections.defaultdict(float)

    def step(it):
        t0 = time.perf_counter()
        sched.step(it)
        opt.zero_grad()
        loss, tb, _ = fn(model, dict(batches[it % 2]))
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        opt.step()
        t3 = time.perf_counter()
        phase["forward+loss"] += t1 - t0
        phase["backward"] += t2 - t1
        phase["optimizer"] += t3 - t2

    for it in range(5):
        step(it)
    torch.cuda.synchronize()
    # ---- plain timing first (no wrappers)
    phase.clear()
    t0 = time.perf_counter()
    for it in range(5, 5 + n):
        step(it)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"B = {batch}: {n} steps, host loop {host / n * 1e3:.2f} ms/step, with the final sync {wall / n * 1e3:.2f} ms/step; "
          + "  ".join(f"{k} {v / n * 1e3:.2f}" for k, v in phase.items()))

    # ---- wrap
    L = native.lib()
    for name in dir(L):
        if name.startswith("rd_"):
            setattr(L, name, timed("C   " + name, getattr(L, name)))
    for name, f in list(vars(K).items()):
        if inspect.isfunction(f) and f.__module__ == K.__name__ and not name.startswith("_"):
            setattr(K, name, timed("K   " + name, f))
    for name, cls in list(vars(A).items()):
        if inspect.isclass(cls) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
            for m in ("forward", "backward"):
                if m in vars(cls):
                    setattr(cls, m, staticmethod(timed(f"F   {name}.{m}", vars(cls)[m].__func__)))
    for name in ("param_grad_stream", "operand_weight_split", "defer_weight_layout", "kernel_weight", "split_activation", "zeros_accum", "zeros_stats",
                 "_side_ok", "begin_step", "end_forward", "nchw_to_rows", "rows_to_nchw", "dense_conv_spec", "linear_spec", "bn_eval_scale_shift",
                 "conv_inference", "conv_bn_act_train"):
        if hasattr(A, name):
            setattr(A, name, timed("A   " + name, getattr(A, name)))
    for name in ("empty", "empty_like", "zeros", "cat"):
        setattr(torch, name, timed("T   torch." + name, getattr(torch, name)))
    for name in ("record_stream", "contiguous", "permute", "reshape", "view"):
        pass        # TensorBase methods cannot be patched; see aten_callers.py for those
    phase.clear()
    ACC.clear()
    t0 = time.perf_counter()
    for it in range(5 + n, 5 + 2 * n):
        step(it)
    host2 = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"with wrappers: host loop {host2 / n * 1e3:.2f} ms/step; " + "  ".join(f"{k} {v / n * 1e3:.2f}" for k, v in phase.items()))
    for prefix, title in (("C   ", "C-ABI calls (inside the library)"), ("K   ", "kernels.py wrappers (inclusive)"), ("A   ", "autograd.py helpers (inclusive)"),
                          ("F   ", "autograd Functions (inclusive)"), ("T   ", "torch allocation / cat")):
        rows = [(k, v) for k, v in ACC.items() if k.startswith(prefix)]
        tot = sum(v[1] for _, v in rows)
        cnt = sum(v[0] for _, v in rows)
        print(f"---- {title}: {tot / n * 1e3:.2f} ms/step in {cnt / n:.0f} calls/step")
        for k, v in sorted(rows, key=lambda kv: -kv[1][1])[:25]:
            print(f"   {k[4:]:44s} {v[0] / n:7.1f} calls  {v[1] / n * 1e3:7.3f} ms/step  {v[1] / max(v[0], 1) * 1e6:6.1f} us/call")


if __name__ == "__main__":
    main()
