#!/usr/bin/env python
"""BASELINE configs[4] sweep: DenseEnc (A8 graph) forward at the 1024 x 1024 BEV shapes -- x_conv4 (B, 256, 128, 128), x_conv5
(B, 256, 64, 64), B in {1, 8} -- in fp32-class (bf16x3), bf16 and fp8 storage.  One JSON line per (precision, B): time per forward,
algorithmic TFLOP/s against the dense MFMA peak of the operand type, algorithmic HBM bytes and the GB/s they imply.

    python tools/bench_lowp.py [--reps 20]
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_lowp -- python tools/bench_lowp.py          (kernel durations)
    rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_... -- python tools/bench_lowp.py --only fp8 --batch 8   (counters, own pass)
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PEAK = {"bf16x3": 2500.0, "bf16": 2500.0, "fp8": 5000.0}          # dense MFMA TFLOP/s of the operand type (MI355X_MICROARCH.md)
ELT = {"bf16x3": 4, "bf16": 2, "fp8": 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--batch", type=int, default=0)
    args = ap.parse_args()
    from radardistill_amd import kernels as K, lowp as LP
    from radardistill_amd.pcdet.config import AttrDict
    from radardistill_amd.pcdet.models.backbones_2d import __all__ as REG
    dev = "cuda:0"
    cfg = dict(LAYER_NUMS=[5, 5], LAYER_STRIDES=[1, 2], NUM_FILTERS=[256, 256], UPSAMPLE_STRIDES=[1, 2], NUM_UPSAMPLE_FILTERS=[128, 128])
    torch.manual_seed(0)
    m = REG["BaseBEVBackboneV2"](AttrDict(cfg), input_channels=256).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    for name, buf in m.named_buffers():
        if name.endswith("running_mean"):
            buf.copy_(torch.randn(buf.shape, generator=g) * 0.1)
        elif name.endswith("running_var"):
            buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
    for p in m.parameters():
        p.requires_grad_(False)
    for B in ([args.batch] if args.batch else [1, 8]):
        rng = np.random.default_rng(B)
        x4 = torch.from_numpy(rng.normal(size=(B, 256, 128, 128)).astype(np.float32)) * torch.from_numpy((rng.uniform(size=(B, 1, 128, 128)) < 0.4).astype(np.float32))
        x5 = torch.from_numpy(rng.normal(size=(B, 256, 64, 64)).astype(np.float32))
        x4 = x4.to(dev).contiguous(memory_format=torch.channels_last)
        x5 = x5.to(dev).contiguous(memory_format=torch.channels_last)
        flops = 2.0 * 9 * 256 * 256 * B * (128 * 128 * 5 + 64 * 64 * 6) + 2.0 * 9 * 512 * 256 * B * 128 * 128 + 2.0 * 4 * 256 * 256 * B * 64 * 64
        # algorithmic bytes: every layer reads its input and writes its output once, weights once
        px4, px5 = B * 128 * 128, B * 64 * 64
        acts = px5 * 256 * 2 * 6 + (px5 * 256 + px4 * 256) + (px4 * 512 + px4 * 256) + px4 * 256 * 2 * 5
        wts = 9 * 256 * 256 * 11 + 9 * 512 * 256 + 4 * 256 * 256
        for prec in ("bf16x3", "bf16", "fp8"):
            if args.only and prec != args.only:
                continue
            if prec == "bf16x3":
                K.set_conv_math("bf16x3")
                fn = lambda: m.dense_enc(x4, x5)
            else:
                eng = LP.LowpDenseEnc(m, LP.BF16 if prec == "bf16" else LP.FP8)
                eng.calibrate(x4, x5)
                fn = lambda: eng.forward(x4, x5)
            with torch.no_grad():
                for _ in range(3):
                    fn()
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
                for _ in range(args.reps):
                    fn()
                t1.record()
                torch.cuda.synchronize()
            ms = t0.elapsed_time(t1) / args.reps
            byts = (acts + wts) * ELT[prec]
            print(json.dumps({"workload": "DenseEnc forward, 1024x1024 BEV (BASELINE configs[4])", "precision": prec, "batch": B, "ms": round(ms, 4),
                              "samples_per_s": round(B / ms * 1e3, 1), "algorithmic_tflops": round(flops / ms / 1e9, 1),
                              "mfma_peak_tflops": PEAK[prec], "frac_of_mfma_peak": round(flops / ms / 1e9 / PEAK[prec], 4),
                              "algorithmic_hbm_mb": round(byts / 1e6, 1), "implied_gbs": round(byts / ms / 1e6, 1),
                              "frac_of_hbm_peak": round(byts / ms / 1e6 / 8000.0, 4)}), flush=True)
            K.set_conv_math("f32")


if __name__ == "__main__":
    main()
