"""Diagnostic (GPU box): host enqueue time of the training forward by top-level module of the detector's module_list (+ the loss), no
device sync inside the step, host-bound batch by default.  Says where the ~5 ms of forward host time outside the autograd Functions go.

    python tools/diag/forward_modules.py [batch=1] [steps=40]
        RD_DP_REHEARSE=1            the data-parallel path in a world of one rank (hooks, bucket launches, RCCL calls), optimizer methods timed
        RD_DIAG_TRAINING_STREAM=1   the loop on a high-priority stream, as bench.py runs it
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
    if os.environ.get("RD_DP_REHEARSE") == "1":          # the data-parallel path in a world of one rank (dist.rehearsal)
        from radardistill_amd import dist as D
        D.init_distributed(backend="nccl", device=device)
        model = D.data_parallel(model, opt, 0)
        for name in ("_grad_ready", "_launch_bucket", "zero_grad", "step", "allreduce_gradients", "_presence", "_fill_table"):
            setattr(opt, name, timed("optimizer." + name, getattr(opt, name)))
        A._flush_deferred_layouts = timed("_flush_deferred_layouts", A._flush_deferred_layouts)
        A.deliver_grads = timed("deliver_grads", A.deliver_grads)
    batches = [B.device_batch(make_batch(batch_size=batch, n_lidar=35000, n_radar=2000, n_boxes=30, grid=512, seed=i), device) for i in range(2)]

    if os.environ.get("RD_DIAG_TRAINING_STREAM") == "1":          # as bench.py: the loop on a high-priority stream
        from radardistill_amd.train import use_training_stream
        use_training_stream(device, -1)
    phase = [0.0, 0.0, 0.0]

    def step(it):          # (phases timed on the host, no device sync)
        t0 = time.perf_counter()
        sched.step(it)
        opt.zero_grad()
        loss, tb, _ = fn(model, dict(batches[it % 2]))
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        opt.step()
        t3 = time.perf_counter()
        phase[0] += t1 - t0; phase[1] += t2 - t1; phase[2] += t3 - t2

    for it in range(6):
        step(it)
    torch.cuda.synchronize()
    phase[:] = [0.0, 0.0, 0.0]
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
    print(f"B = {batch}: {n} steps, host loop {host / n * 1e3:.2f} ms/step; forward+loss {phase[0] / n * 1e3:.2f}  backward {phase[1] / n * 1e3:.2f}  "
          f"optimizer {phase[2] / n * 1e3:.2f}")
    # CPU time of every thread of this process (a busy helper thread -- a collective library's proxy or watchdog -- competes with the
    # enqueue loop for the box's CPU share)
    tick = os.sysconf("SC_CLK_TCK")
    threads = []
    for tid in os.listdir("/proc/self/task"):
        try:
            comm = open(f"/proc/self/task/{tid}/comm").read().strip()
            f = open(f"/proc/self/task/{tid}/stat").read().rsplit(")", 1)[1].split()
            threads.append(((int(f[11]) + int(f[12])) / tick, comm, tid))
        except OSError:
            pass
    print("threads by CPU seconds:", ", ".join(f"{c} {t:.2f}s" for t, c, _ in sorted(threads, reverse=True)[:10]), f"({len(threads)} threads)")
    for k, (c, t) in sorted(ACC.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:58s} {c / n:6.1f} calls  {t / n * 1e3:8.3f} ms/step")


if __name__ == "__main__":
    main()
