"""Isolated timing of the gathered implicit-GEMM kernel on a submanifold sparse 3x3 convolution (bf16x3): random active sites at a
given occupancy of a B x H x W grid, Cin -> Cout.

    python tools/diag/sparse_micro.py [B H W occupancy Cin Cout] [--iters N]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from radardistill_amd import kernels as K, sparse as SP      # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    iters = 30
    for i, a in enumerate(sys.argv):
        if a == "--iters":
            iters = int(sys.argv[i + 1]); args.remove(sys.argv[i + 1])
    B, H, W, occ, Cin, Cout = (int(args[0]), int(args[1]), int(args[2]), float(args[3]), int(args[4]), int(args[5])) if len(args) == 6 else (8, 64, 64, 0.52, 256, 256)
    dev = torch.device("cuda:0")
    K.set_conv_math(os.environ.get("RD_MATH", "bf16x3"))
    rng = np.random.default_rng(3)
    n = int(B * H * W * occ)
    keys = np.sort(rng.choice(B * H * W, size=n, replace=False))
    idx = np.stack([keys // (H * W), (keys // W) % H, keys % W], axis=1).astype(np.int32)
    t = SP.SparseConvTensor(torch.randn(n, Cin, device=dev), torch.from_numpy(idx).to(dev), [H, W], B)
    spec = t._level.subm_spec()
    g = torch.Generator(device="cpu").manual_seed(1)
    w = (torch.randn(Cout, 9, Cin, generator=g) / (9 * Cin) ** 0.5).to(dev)
    ws = int(os.environ.get("RD_WS", "1")) if (K.get_conv_math() == "bf16x3" and Cout > 32) else 0          # 2: fragment-major (k_gemm_b3f<.., TABLE>)
    if ws:
        w = K.weight_layout_split(w, Cout, Cin, 9, 0, frag=ws == 2)
    x = t.features
    pairs = int((spec.fwd_nbr >= 0).sum())

    def run():
        return K.conv_fwd(x, w, 9, None, n, Cout, spec.fwd_ix, nbr_keepalive=spec.fwd_nbr, w_split=ws)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * pairs * Cin * Cout
    print(f"subm {B}x{H}x{W} occ {occ}: {n} rows, {pairs / n:.2f} neighbours/row, {Cin}->{Cout}: {ms * 1e3:.1f} us/launch  {fl / ms / 1e9:.1f} TF/s algorithmic  math={K.get_conv_math()}")


if __name__ == "__main__":
    main()
