"""Isolated timing of one dense 3x3 stride-1 weight gradient (bf16x3 mode): k_conv_wgrad_d3_b3 (halo-staged; RD_WGRAD_D3=0 selects the
gathered transposing-read kernel k_conv_wgrad_tr_b3).  RD_WGRAD_D3_RES=256 lifts the 80-CU share the step runs it with.

    python tools/diag/wgrad_micro.py [B H W Cin Cout] [--iters N]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from radardistill_amd import autograd as A, kernels as K      # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    iters = 20
    for i, a in enumerate(sys.argv):
        if a == "--iters":
            iters = int(sys.argv[i + 1]); args.remove(sys.argv[i + 1])
    B, H, W, Cin, Cout = [int(v) for v in args] if len(args) == 5 else (8, 64, 64, 256, 256)
    dev = torch.device("cuda:0")
    K.set_conv_math(os.environ.get("RD_MATH", "bf16x3"))
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(B * H * W, Cin, generator=g).to(dev)
    go = torch.randn(B * H * W, Cout, generator=g).to(dev)
    spec = A.dense_conv_spec(B, H, W, 3, 3, 1, 1)
    A.begin_step(dev)
    for _ in range(3):
        K.conv_wgrad(x, go, 9, spec.fwd_ix)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    A.begin_step(dev)
    e0.record()
    for _ in range(iters):
        K.conv_wgrad(x, go, 9, spec.fwd_ix)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * B * H * W * 9 * Cin * Cout
    print(f"wgrad {B}x{H}x{W} {Cin}->{Cout} 3x3: {ms:.4f} ms/launch  {fl / ms / 1e9:.1f} TF/s algorithmic  (incl. the accumulator fill)  "
          f"RD_WGRAD_D3={os.environ.get('RD_WGRAD_D3', '1')} RES={os.environ.get('RD_WGRAD_D3_RES', '80')}")


if __name__ == "__main__":
    main()
