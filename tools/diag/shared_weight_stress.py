"""Stress form of tests/test_gpu_model.py::test_shared_weight_and_hooked_weight_gradients_with_deferred_layout: many repetitions in one
process, reporting which gradient (shared weight / hooked weight / input) deviates and by how much."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from radardistill_amd import autograd as A, kernels as K

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    hook = (sys.argv[2] != "nohook") if len(sys.argv) > 2 else True
    math = sys.argv[3] if len(sys.argv) > 3 else "f32"
    g = np.random.default_rng(11)
    B, H, W, C = 2, 12, 16, 64
    x = torch.from_numpy(g.normal(size=(B, C, H, W)).astype(np.float32))
    w = torch.from_numpy((g.normal(size=(C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32))
    w2 = torch.from_numpy((g.normal(size=(C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32))
    wr, w2r = w.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    if hook:
        w2r.register_hook(lambda gr: gr * 2.0)
    F.conv2d(F.conv2d(F.conv2d(xr, wr, None, 1, 1), w2r, None, 1, 1), wr, None, 1, 1).square().sum().backward()
    K.set_conv_math(math)
    dev = torch.device("cuda", 0)
    bad = [0, 0, 0]
    junk = []
    for it in range(n):
        spec = A.dense_conv_spec(B, H, W, 3, 3, 1, 1)
        rows = x.permute(0, 2, 3, 1).reshape(-1, C).contiguous().to(dev).requires_grad_(True)
        wd, w2d = torch.nn.Parameter(w.to(dev)), torch.nn.Parameter(w2.to(dev))
        if hook:
            w2d.register_hook(lambda gr: gr * 2.0)
        junk = [torch.randn(1 << 20, device=dev) for _ in range(it % 4)]          # perturb the allocator / queue
        A.begin_step(dev)
        y = A.conv(A.conv(A.conv(rows, wd, None, spec, C), w2d, None, spec, C), wd, None, spec, C)
        y.square().sum().backward()
        torch.cuda.synchronize()
        for i, (got, ref) in enumerate(((wd.grad, wr.grad), (w2d.grad, w2r.grad), (rows.grad.view(B, H, W, C).permute(0, 3, 1, 2), xr.grad))):
            err = float((got.cpu() - ref).abs().max()) / float(ref.abs().max())
            if err > 1e-3:
                bad[i] += 1
                print(f"iter {it}: tensor {i} rel err {err:.3e}", flush=True)
    print(f"math {math} hook {hook}: {n} iterations, deviations shared/hooked/input = {bad}  WGRAD_STREAM={A.WGRAD_STREAM[0]}", flush=True)

main()
