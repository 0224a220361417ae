"""Does one deterministic training step depend on what freshly allocated device memory happens to contain?
    python tools/diag/poison_step.py [none|nan|big|zero] [f32|bf16x3]   -> prints a hash of the loss and of every gradient
Allocates most of a 4-GiB arena filled with the poison value and frees it again (the caching allocator then hands those bytes out
for the step's torch.empty buffers), runs ONE step in deterministic mode, prints sha1 over all gradients."""
import hashlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from radardistill_amd import kernels as K
from radardistill_amd.pcdet.models import model_fn_decorator
from radardistill_amd.synthetic import make_batch
from tests.test_gpu_model import _build_pillarnet
from tests.seeded import seeded_fill_

mode = sys.argv[1] if len(sys.argv) > 1 else "none"
math = sys.argv[2] if len(sys.argv) > 2 else "f32"
K.set_conv_math(math)
model, cfg, pc_range, voxel, gs = _build_pillarnet(128)
sd = model.state_dict(); seeded_fill_(sd, seed=77); model.load_state_dict(sd)
model = model.to("cuda").train()
batch = make_batch(batch_size=2, n_lidar=300, n_radar=700, n_boxes=10, grid=128, seed=5)
if mode != "none":
    val = {"nan": float("nan"), "big": 3.0e30, "zero": 0.0}[mode]
    junk = [torch.full((s,), val, device="cuda") for s in (1 << 28, 1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14, 1 << 12)] + \
           [torch.full((n,), val, device="cuda") for n in [3000] * 200 + [70000] * 100 + [1 << 20] * 50]
    torch.cuda.synchronize()
    del junk
K.set_deterministic(True)
loss, tb, _ = model_fn_decorator()(model, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()})
loss.backward()
torch.cuda.synchronize()
h = hashlib.sha1()
bad = []
for k, p in sorted(model.named_parameters()):
    if p.grad is not None:
        g = p.grad.detach().cpu().numpy()
        if not np.isfinite(g).all():
            bad.append(k)
        h.update(g.tobytes())
print(f"poison={mode} math={math} loss={float(loss):.9f} grads sha1={h.hexdigest()[:16]} nonfinite={bad[:5]} ({len(bad)})", flush=True)
