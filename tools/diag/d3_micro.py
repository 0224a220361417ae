"""Isolated timing of one dense 3x3 stride-1 convolution (bf16x3 mode) for PMC collection / kernel tuning.

    python tools/diag/d3_micro.py [B H W Cin Cout] [--iters N]        (RD_D3=0 selects the gathered kernel instead of the halo kernel)
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
    w = (torch.randn(Cout, 9, Cin, generator=g) / (9 * Cin) ** 0.5).to(dev)
    spec = A.dense_conv_spec(B, H, W, 3, 3, 1, 1)
    ws = int(os.environ.get("RD_WS", "1"))          # 0: fp32 weights, 1: split format (LDS-staged kernel), 2: fragment-major (k_conv_d3f_b3)
    if ws:
        w = K.weight_layout_split(w, Cout, Cin, 9, 0, frag=ws == 2)
    for _ in range(3):
        K.conv_fwd(x, w, 9, None, B * H * W, Cout, spec.fwd_ix, w_split=ws)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        K.conv_fwd(x, w, 9, None, B * H * W, Cout, spec.fwd_ix, w_split=ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * B * H * W * 9 * Cin * Cout
    print(f"conv {B}x{H}x{W} {Cin}->{Cout} 3x3: {ms:.4f} ms/launch  {fl / ms / 1e9:.1f} TF/s algorithmic  (x3 = {3 * fl / ms / 1e9:.1f} bf16 TF/s)  RD_D3={os.environ.get('RD_D3', '1')} w_split={ws}")


if __name__ == "__main__":
    main()
