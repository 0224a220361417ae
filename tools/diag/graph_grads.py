"""Diagnostic (GPU box): per-parameter gradient differences eager vs HIP-graphed at step 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from radardistill_amd.pcdet.models import model_fn_decorator
from radardistill_amd.synthetic import make_batch
from tests.seeded import seeded_fill_
from tests.test_gpu_model import _build_pillarnet, DEV

def run(graphs):
    model, cfg, pc_range, voxel, gs = _build_pillarnet(128)
    sd = model.state_dict(); seeded_fill_(sd, seed=78); model.load_state_dict(sd)
    model = model.to(DEV); model.train(); model.use_graphs = graphs
    batch = make_batch(batch_size=2, n_lidar=300, n_radar=700, n_boxes=10, grid=128, seed=50)
    loss, tb, _ = model_fn_decorator()(model, dict(batch))
    loss.backward()
    return {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in model.named_parameters() if p.requires_grad}, float(loss)

ge, le = run(False)
gg, lg = run(True)
print("loss", le, lg)
gmax = max(float(v.norm()) for v in ge.values() if v is not None)
print("total grad norm eager", float(torch.sqrt(sum((v.double() ** 2).sum() for v in ge.values() if v is not None))),
      "graph", float(torch.sqrt(sum((v.double() ** 2).sum() for v in gg.values() if v is not None))))
rows = []
for k in ge:
    a, b = ge[k], gg[k]
    if a is None or b is None:
        rows.append((float("inf"), k, a is None, b is None)); continue
    rows.append((float((a - b).norm() / (a.norm() + 1e-3 * gmax)), k, float(a.norm()), float(b.norm())))
rows.sort(reverse=True, key=lambda r: r[0])
for r in rows[:14]:
    print(r)
