"""Diagnostic (GPU box): run-to-run divergence of 3 training steps, eager vs eager vs HIP-graphed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from radardistill_amd.pcdet.models import model_fn_decorator
from radardistill_amd.synthetic import make_batch
from radardistill_amd.train import build_optimizer, build_scheduler
from tests.seeded import seeded_fill_
from tests.test_gpu_model import _build_pillarnet, DEV

def run(graphs):
    model, cfg, pc_range, voxel, gs = _build_pillarnet(128)
    sd = model.state_dict(); seeded_fill_(sd, seed=78); model.load_state_dict(sd)
    model = model.to(DEV); model.train(); model.use_graphs = graphs
    opt = build_optimizer(model, cfg.OPTIMIZATION)
    sched, _ = build_scheduler(opt, 100, 1, -1, cfg.OPTIMIZATION)
    out = []
    for it in range(3):
        batch = make_batch(batch_size=2, n_lidar=300, n_radar=700, n_boxes=10, grid=128, seed=50 + it)
        sched.step(it); opt.zero_grad()
        loss, tb, _ = model_fn_decorator()(model, dict(batch))
        loss.backward(); norm = opt.step()
        out.append(dict(loss=float(loss), gnorm=float(norm[0]), **{k: float(v) for k, v in tb.items()}))
    return out

a, b, g = run(False), run(False), run(True)
for it in range(3):
    print("step", it, "loss eager/eager/graph", a[it]["loss"], b[it]["loss"], g[it]["loss"], "gnorm", a[it]["gnorm"], b[it]["gnorm"], g[it]["gnorm"])
    worst_ee = max((abs(a[it][k] - b[it][k]) / (abs(a[it][k]) + 1e-6), k) for k in a[it])
    worst_eg = max((abs(a[it][k] - g[it][k]) / (abs(a[it][k]) + 1e-6), k) for k in a[it])
    print("   worst rel dev eager-vs-eager", worst_ee, " eager-vs-graph", worst_eg)
