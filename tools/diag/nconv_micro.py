"""Isolated timing of the CenterHead's batched narrow convolutions (nconv.hip) at the step's shape: 42 branches of 64 channels over
8 x 64 x 64 pixels (y = 352 MB), 76 output columns.  Prints us / launch and the algorithmic streaming rate of y for forward, weight gradient
and the fused BatchNorm-backward data gradient.

    python tools/diag/nconv_micro.py [B H W]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from radardistill_amd import autograd as A, kernels as K      # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    args = [int(v) for v in sys.argv[1:]]
    B, H, W = args if len(args) == 3 else (8, 64, 64)
    dev = torch.device("cuda:0")
    widths = [1, 2, 1, 3, 2, 2, 1] * 6          # hm / center / center_z / dim / rot / vel / iou of the 6 task heads (hm 1-2)
    cols, c = [], 0
    for w_ in widths:
        cols.append(c); c += w_
    tab = K.BranchTable([64 * i for i in range(len(widths))], cols, widths)
    rows = B * H * W
    g = torch.Generator(device="cpu").manual_seed(1)
    y = torch.randn(rows, 64 * len(widths), device=dev)
    w = torch.randn(tab.no, 64, 3, 3, device=dev) * 0.05
    bias = torch.randn(tab.no, device=dev)
    go = torch.randn(rows, tab.no, device=dev)
    side = torch.rand(4, y.shape[1], device=dev) + 0.5
    gamma = torch.rand(y.shape[1], device=dev) + 0.5
    mb = y.numel() * 4 / 1e6
    A.begin_step(dev)
    t = timed(lambda: K.nconv_fwd(y, w, bias, B, H, W, tab))
    print(f"nconv_fwd      {t:8.1f} us   {mb / t:6.2f} TB/s of y ({mb:.0f} MB)")
    t = timed(lambda: (A.begin_step(dev), K.nconv_wgrad(y, go, B, H, W, tab)))
    print(f"nconv_wgrad    {t:8.1f} us   {mb / t:6.2f} TB/s of y (incl. the arena memset of begin_step)")
    t = timed(lambda: (A.begin_step(dev), K.nconv_dgrad_bn(go, w, y, gamma, side, B, H, W, tab)))
    print(f"nconv_dgrad_bn {t:8.1f} us   {3 * mb / t:6.2f} TB/s (x read twice, grad_x written once)")


if __name__ == "__main__":
    main()
