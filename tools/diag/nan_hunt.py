"""Diagnostic (GPU box): run the bench workload for N steps and report when a loss term first becomes non-finite, and which one.

    python tools/diag/nan_hunt.py [--steps 400] [--fused 1|0] [--math bf16x3|f32] [--total 405]
"""
import argparse
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--fused", type=int, default=1)
    ap.add_argument("--math", default="bf16x3")
    ap.add_argument("--total", type=int, default=405)
    ap.add_argument("--every", type=int, default=25)
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    from radardistill_amd import kernels as K
    from radardistill_amd.pcdet.models import model_fn_decorator
    from radardistill_amd.synthetic import make_batch
    from radardistill_amd.train import build_optimizer, build_scheduler
    K.set_conv_math(args.math)
    model, cfg, geom = B.build(os.path.join(ROOT, "tools/cfgs/radar_distill/bench_512.yaml"), 512, device)
    for m in model.modules():
        if hasattr(m, "model_cfg") and hasattr(m.model_cfg, "get") and m.model_cfg.get("LOSS_CONFIG", None) is not None:
            m.model_cfg.LOSS_CONFIG["FUSED"] = bool(args.fused)
    model.train()
    opt = build_optimizer(model, cfg.OPTIMIZATION)
    sched, _ = build_scheduler(opt, max(args.total, 10), 1, -1, cfg.OPTIMIZATION)
    fn = model_fn_decorator()
    batches = [B.device_batch(make_batch(batch_size=8, n_lidar=35000, n_radar=2000, n_boxes=30, grid=512, seed=i), device) for i in range(2)]
    first_bad = None
    for it in range(args.steps):
        sched.step(it)
        opt.zero_grad()
        loss, tb, _ = fn(model, dict(batches[it % 2]))
        loss.backward()
        norm = opt.step()
        if it % args.every == 0 or it == args.steps - 1 or first_bad is None:
            vals = {k: float(v) for k, v in tb.items()}
            bad = [k for k, v in vals.items() if not math.isfinite(v)]
            gn = [float(x) for x in norm.tolist()] if norm is not None else None
            if it % args.every == 0 or it == args.steps - 1:
                print(f"step {it}: loss {float(loss):.4f} lr {opt.lr:.2e} gradnorm/clip {gn} feature {vals.get('loss_feature', float('nan')):.4f} rpn {vals.get('rpn_loss', float('nan')):.4f}", flush=True)
            if bad and first_bad is None:
                first_bad = it
                print(f"FIRST NON-FINITE at step {it}: {bad[:12]}  lr {opt.lr:.3e} gradnorm/clip {gn}", flush=True)
                print({k: round(v, 4) for k, v in vals.items() if 'head_0' in k or 'feature' in k or 'low' in k or 'high' in k or 'mask' in k}, flush=True)
    print("done; first non-finite step:", first_bad)


if __name__ == "__main__":
    main()
