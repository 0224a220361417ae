"""shared_weight_stress.py after a full training step in the same process (arena sizes, streams and allocator pools as inside the suite)."""
import os, sys, runpy
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from radardistill_amd import autograd as A, kernels as K
from radardistill_amd.pcdet.models import model_fn_decorator
from radardistill_amd.synthetic import make_batch
from tests.test_gpu_model import _build_pillarnet
from tests.seeded import seeded_fill_

model, cfg, pc_range, voxel, gs = _build_pillarnet(128)
sd = model.state_dict(); seeded_fill_(sd, seed=77); model.load_state_dict(sd)
model = model.to("cuda").train()
batch = make_batch(batch_size=2, n_lidar=300, n_radar=700, n_boxes=10, grid=128, seed=5)
fn = model_fn_decorator()
for det in (True, False):
    K.set_deterministic(det)
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        loss, _, _ = fn(model, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()})
        loss.backward()
    torch.cuda.synchronize()
K.set_deterministic(False)
if len(sys.argv) > 4 and sys.argv[4] == "keep":
    pass
else:
    del model
sys.argv = [sys.argv[0]] + sys.argv[1:4]
runpy.run_path(os.path.join(ROOT, "tools/diag/shared_weight_stress.py"), run_name="__main__")
