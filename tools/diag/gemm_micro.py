"""Isolated timing of the gathered implicit-GEMM kernel on plain GEMM shapes (1x1 convolution / nn.Linear): rows x Cin -> Cout.

    python tools/diag/gemm_micro.py [rows Cin Cout] [--iters N]
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
    rows, Cin, Cout = [int(v) for v in args] if len(args) == 3 else (8192, 256, 1024)
    dev = torch.device("cuda:0")
    K.set_conv_math(os.environ.get("RD_MATH", "bf16x3"))
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(rows, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, 1, Cin, generator=g) / Cin ** 0.5).to(dev)
    ws = int(os.environ.get("RD_WS", "1")) if K.get_conv_math() == "bf16x3" else 0          # 1: split format (gathered kernel), 2: fragment-major (k_gemm_b3f)
    if ws:
        w = K.weight_layout_split(w, Cout, Cin, 1, 0, frag=ws == 2)
    spec = A.linear_spec(rows)
    for _ in range(3):
        K.conv_fwd(x, w, 1, None, rows, Cout, spec.fwd_ix, w_split=ws)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        K.conv_fwd(x, w, 1, None, rows, Cout, spec.fwd_ix, w_split=ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * rows * Cin * Cout
    by = 4.0 * (rows * Cin + Cin * Cout + rows * Cout)
    print(f"gemm {rows}x{Cin}->{Cout}: {ms * 1e3:.1f} us/launch  {fl / ms / 1e9:.1f} TF/s algorithmic  {by / ms / 1e6:.0f} GB/s algorithmic  math={K.get_conv_math()}")
    if "--floors" in sys.argv:
        # reference points for the same shape (diagnostic only): the BLAS library's plain GEMMs and a pure stream of the output size
        def timed(fn, n=iters):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) / n * 1e3
        wt = torch.randn(Cin, Cout, device=dev)
        out = torch.empty(rows, Cout, device=dev)
        print(f"  torch.mm fp32            {timed(lambda: torch.mm(x, wt, out=out)):.1f} us")
        xb, wb, ob = x.bfloat16(), wt.bfloat16(), out.bfloat16()
        print(f"  torch.mm bf16 (bf16 out) {timed(lambda: torch.mm(xb, wb, out=ob)):.1f} us")
        y = torch.empty(rows, Cout, device=dev)
        print(f"  rd_affine_act on the output ({rows}x{Cout}: read + write) {timed(lambda: K.affine_act(y, None, None, None, 1)):.1f} us")
        print(f"  out.zero_() ({rows * Cout * 4 / 1e6:.0f} MB write) {timed(lambda: out.zero_()):.1f} us")


if __name__ == "__main__":
    main()
