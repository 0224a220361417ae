"""Is one deterministic training step the same at the start of a process and after other work in the same process?"""
import hashlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from radardistill_amd import kernels as K, autograd as A
from radardistill_amd.pcdet.models import model_fn_decorator
from radardistill_amd.synthetic import make_batch
import tests.test_gpu_model as TM
import tests.test_gpu_kernels as TK
from tests.seeded import seeded_fill_

def one(tag):
    model, cfg, pc_range, voxel, gs = TM._build_pillarnet(128)
    sd = model.state_dict(); seeded_fill_(sd, seed=77); model.load_state_dict(sd)
    model = model.to("cuda").train()
    batch = make_batch(batch_size=2, n_lidar=300, n_radar=700, n_boxes=10, grid=128, seed=5)
    K.set_deterministic(True)
    try:
        loss, tb, _ = model_fn_decorator()(model, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in batch.items()})
        loss.backward()
        torch.cuda.synchronize()
    finally:
        K.set_deterministic(False)
    h = hashlib.sha1()
    per = {}
    for k, p in sorted(model.named_parameters()):
        if p.grad is not None:
            b = p.grad.detach().cpu().numpy().tobytes()
            h.update(b); per[k] = hashlib.sha1(b).hexdigest()[:8]
    print(f"{tag}: loss={float(loss):.9f} grads sha1={h.hexdigest()[:16]}", flush=True)
    return per

a = one("fresh")
b = one("second model, same process")
TK.test_sparse_enc_c2_vs_oracle(True)
TK.test_sparse_conv_forward_and_backward(64, 128)
c = one("after sparse tests")
for name in ("test_batchnorm_train_forward_backward",):
    getattr(TK, name)(256, 777, 1, False)
TK.test_fused_conv_bn_act_node_equals_separate_nodes(1, True)
d = one("after bn / fused node tests")
for x, y, t in ((a, b, "fresh vs second"), (a, c, "fresh vs after sparse"), (a, d, "fresh vs after bn")):
    diff = [k for k in x if x[k] != y[k]]
    print(t, "differing tensors:", len(diff), diff[:6])
