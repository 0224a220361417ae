"""Diagnostic (GPU box): the same training step (test geometry: 128x128 BEV, B=2) under every combination of the stream features
(geometry prelude, weight-gradient side stream, teacher stream); all gradients compared with the all-off run.  Any tensor that moves
by more than atomics noise points at a cross-stream ordering / memory-lifetime bug."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from radardistill_amd import autograd as A                                   # noqa: E402
from radardistill_amd.pcdet.models import model_fn_decorator                  # noqa: E402
from radardistill_amd.synthetic import make_batch                             # noqa: E402
import tests.test_gpu_model as T                                              # noqa: E402
from tests.seeded import seeded_fill_                                         # noqa: E402


def main():
    grid, B = 128, 2
    model, cfg, pc_range, voxel, gs = T._build_pillarnet(grid)
    sd = model.state_dict(); seeded_fill_(sd, seed=77); model.load_state_dict(sd)
    model = model.to("cuda:0")
    batch = make_batch(batch_size=B, n_lidar=300, n_radar=700, n_boxes=10, grid=grid, seed=5)
    fn = model_fn_decorator()
    ref = None
    for geom, wgrad, teacher in ((0, 0, 0), (0, 1, 0), (1, 0, 0), (1, 1, 0)):
        os.environ["RD_GEOM_STREAM"] = str(geom)
        os.environ["RD_TEACHER_STREAM"] = str(teacher)
        A.WGRAD_STREAM[0] = bool(wgrad)
        model.train()
        model.zero_grad(set_to_none=True)
        loss, tb, _ = fn(model, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()})
        loss.backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        if ref is None:
            ref = grads
            print("reference run: loss", float(loss))
            continue
        bad = []
        for k, g in grads.items():
            rel = float((g - ref[k]).norm() / ref[k].norm().clamp_min(1e-12))
            if rel > 2e-2:
                bad.append((rel, k))
        print(f"geom={geom} wgrad={wgrad} teacher={teacher}: loss {float(loss):.6f}  tensors off by > 2e-2: {len(bad)}", flush=True)
        for rel, k in [b for b in sorted(bad, reverse=True) if not b[1].endswith('.bias')][:12]:
            print(f"      {rel:9.3e}  {k}")


if __name__ == "__main__":
    main()
