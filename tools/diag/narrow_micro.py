"""Isolated timing of the 256 -> 27 stride-2 DCNv2 offset / mask convolution (bf16x3 mode): split-K wavefront kernel
(conv_small.hip k_conv_narrow_b3) vs the tiled exact-fp32 kernel (RD_CONV_NARROW=0).

    python tools/diag/narrow_micro.py [B H W Cin Cout]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from radardistill_amd import autograd as A, kernels as K      # noqa: E402


def main():
    args = [int(a) for a in sys.argv[1:]]
    B, H, W, Cin, Cout = args if len(args) == 5 else (8, 128, 128, 256, 27)
    dev = torch.device("cuda:0")
    K.set_conv_math("bf16x3")
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(B * H * W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, 9, Cin, generator=g) / (9 * Cin) ** 0.5).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    spec = A.dense_conv_spec(B, H, W, 3, 3, 2, 1)
    for _ in range(3):
        out = K.conv_fwd(x, w, 9, b, spec.out_rows, Cout, spec.fwd_ix)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        out = K.conv_fwd(x, w, 9, b, spec.out_rows, Cout, spec.fwd_ix)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    fl = 2.0 * spec.out_rows * 9 * Cin * Cout
    print(f"conv {B}x{H}x{W} {Cin}->{Cout} 3x3 s2: {ms * 1e3:.1f} us/launch  {fl / ms / 1e9:.1f} TF/s algorithmic  RD_CONV_NARROW={os.environ.get('RD_CONV_NARROW', '1')} "
          f"slices={os.environ.get('RD_NARROW_SLICES', '8')}  checksum {float(out.double().sum()):.6f}")


if __name__ == "__main__":
    main()
