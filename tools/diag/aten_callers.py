"""Diagnostic (GPU box): who calls the small ATen ops from Python during one training step?  Counts torch.zeros / zeros_like /
empty-free ops (clone, contiguous-that-copies, copy_, add, mul, cat ...) by the innermost radardistill_amd caller, main thread and
autograd thread alike.  What the torch profiler counts but this does not see is launched by autograd's own C++ nodes."""
import collections
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B                                                              # noqa: E402

COUNTS = collections.Counter()
ELEMS = collections.Counter()          # elements of the largest tensor argument / result: which call sites move real data
ON = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "radardistill_amd" in fr.filename and "aten_callers" not in fr.filename:
            return f"{fr.filename.split('radardistill_amd/')[-1]}:{fr.lineno}"
    return "?"


def wrap(owner, name, label=None, pred=None):
    orig = getattr(owner, name)

    def f(*a, **k):
        out = orig(*a, **k)
        if ON[0] and (pred is None or pred(*a, **k)):
            key = (label or name, site())
            COUNTS[key] += 1
            flat = [t for x in a for t in (x if isinstance(x, (list, tuple)) else [x])] + [out]
            ELEMS[key] += max([t.numel() for t in flat if torch.is_tensor(t)] or [0])
        return out
    setattr(owner, name, f)


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
    for nm in ("zeros", "zeros_like", "cat", "stack", "where", "full", "ones", "arange", "tensor"):
        wrap(torch, nm)
    T = torch.Tensor
    for nm in ("clone", "copy_", "add", "add_", "mul", "mul_", "div", "sum", "float", "long", "bool", "to", "new_zeros", "zero_", "fill_", "__add__", "__mul__",
               "__sub__", "__truediv__", "__neg__", "__rmul__", "__radd__", "__rsub__", "reshape", "index_add_"):
        wrap(T, nm)
    wrap(T, "contiguous", "contiguous(copy)", pred=lambda self, *a, **k: not self.is_contiguous())
    ON[0] = True
    step(3)
    ON[0] = False
    torch.cuda.synchronize()
    tot = collections.Counter()
    for (op, s), c in COUNTS.items():
        tot[op] += c
    print("totals:", dict(tot))
    for (op, s), c in sorted(COUNTS.items(), key=lambda kv: -kv[1])[:60]:
        print(f"{c:5d}  {op:18s} {s}")
    print("---- by elements moved (ops that touch >= 1e5 elements per call)")
    for key, e in sorted(ELEMS.items(), key=lambda kv: -kv[1])[:60]:
        if e / COUNTS[key] >= 1e5 and key[0] not in ("reshape",):
            print(f"{e / 1e6:9.2f} M elements  {COUNTS[key]:4d} x  {key[0]:18s} {key[1]}")


if __name__ == "__main__":
    main()
