"""Isolated timing of the 7x7 depthwise convolution (forward / data gradient and weight gradient) at the CMA shapes.

    python tools/diag/dwconv_micro.py [B H W C]          RD_DWCONV_TILED=0 selects the plain kernels
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from radardistill_amd import kernels as K      # noqa: E402


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    args = [int(a) for a in sys.argv[1:]]
    B, H, W, C = args if len(args) == 4 else (8, 32, 32, 256)
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(B * H * W, C, generator=g).to(dev)
    go = torch.randn(B * H * W, C, generator=g).to(dev)
    w = torch.randn(49, C, generator=g).to(dev)
    b = torch.randn(C, generator=g).to(dev)
    f = timeit(lambda: K.dwconv_fwd(x, w, b, B, H, W, 7))
    wg = timeit(lambda: K.dwconv_wgrad(x, go, B, H, W, 7))
    mb = B * H * W * C * 4 / 1e6
    print(f"dwconv 7x7 {B}x{H}x{W}x{C}: forward {f:.1f} us ({2 * mb / f:.2f} TB/s of in + out), weight gradient {wg:.1f} us   "
          f"RD_DWCONV_TILED={os.environ.get('RD_DWCONV_TILED', '1')}  checksum {float(K.dwconv_fwd(x, w, b, B, H, W, 7).double().sum()):.4f} "
          f"{float(K.dwconv_wgrad(x, go, B, H, W, 7).double().sum()):.4f}")


if __name__ == "__main__":
    main()
