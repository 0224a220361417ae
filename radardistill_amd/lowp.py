"""Low-precision execution of the frozen DenseEnc (BASELINE configs[2]: bf16, configs[4]: fp8 MFMA).

`LowpDenseEnc` runs a BaseBEVBackboneV2 in eval mode (pcdet/models/backbones_2d/base_bev_backbone.py:285-308) on the kernels of
csrc/lowp.hip: weights and every intermediate activation are STORED as bf16 or OCP fp8 (e4m3fn), products accumulate in fp32 on
the matrix cores, BatchNorm is folded into each layer's epilogue.  fp32 stays the parity mode of the library; this path states
its own tolerances (tests/test_gpu_lowp.py) against the fp32 kernels and the reference-generated fixture g2.

Scales (fp8): per-output-channel weight scales max|w| / 448; one scale per activation tensor, max|x| / 448 measured once on a
calibration batch by running the fp32 module (`calibrate`).  bf16 needs no scales.
"""
import torch
import torch.nn as nn

from . import native
from .kernels import _chk, _p, _stream
from .native import check

BF16, FP8, F32 = 0, 1, 2
_ELT = {BF16: 2, FP8: 1, F32: 4}
_TORCH = {BF16: torch.bfloat16, FP8: torch.uint8, F32: torch.float32}       # fp8 bytes are kept in uint8 storage
FP8_MAX = 448.0


def _narrow_empty(rows, cols, dtype, device):
    return torch.empty((rows, cols), dtype=_TORCH[dtype], device=device)


def lp_cast(x_rows, dtype, mul=1.0, out=None, out_col0=0):
    """fp32 rows (n, C) -> narrow rows; `out` (n, ld) places the result at columns [out_col0, out_col0 + C) (concat by placement)."""
    _chk(x_rows, torch.float32, "lp_cast input", 2)
    n, C = x_rows.shape
    if out is None:
        out = _narrow_empty(n, C, dtype, x_rows.device)
    if out.shape[0] != n or out.dtype != _TORCH[dtype] or not out.is_contiguous():
        raise RuntimeError("lp_cast: destination must be contiguous narrow rows matching the input rows")
    check(native.lib().rd_lp_cast(_p(x_rows), n, C, dtype, float(mul), _p(out), out.shape[1], int(out_col0), _stream()), "rd_lp_cast")
    return out


def lp_uncast(x, dtype, mul=1.0):
    if x.dtype != _TORCH[dtype] or not x.is_contiguous() or not x.is_cuda:
        raise RuntimeError("lp_uncast: contiguous CUDA narrow tensor expected")
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(native.lib().rd_lp_uncast(_p(x), x.numel(), dtype, float(mul), _p(out), _stream()), "rd_lp_uncast")
    return out


def lp_amax(x):
    _chk(x, torch.float32, "lp_amax input")
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    check(native.lib().rd_lp_amax(_p(x), x.numel(), _p(out), _stream()), "rd_lp_amax")
    return out


def lp_quant_weights(w_k, dtype):
    """w_k (Cout, taps, Cin) fp32 kernel layout -> (narrow weights, per-output-channel scale (Cout,))."""
    _chk(w_k, torch.float32, "weights")
    Cout = w_k.shape[0]
    K = w_k.numel() // Cout
    wq = torch.empty((Cout, K), dtype=_TORCH[dtype], device=w_k.device)
    scale = torch.empty(Cout, dtype=torch.float32, device=w_k.device)
    check(native.lib().rd_lp_quant_weights(_p(w_k), Cout, K, dtype, _p(wq), _p(scale), _stream()), "rd_lp_quant_weights")
    return wq, scale


def lp_conv(x, dtype, B, H, W, Cin, wq, ksize, alpha, beta, relu, out_dtype, Cout, out=None, out_col0=0, deconv=False):
    """x: narrow rows (B*H*W, ld >= Cin).  -> narrow / fp32 rows (B*Ho*Wo, Cout) (or written into `out` at out_col0)."""
    if x.dtype != _TORCH[dtype] or not x.is_contiguous() or x.shape[0] != B * H * W or x.shape[1] < Cin:
        raise RuntimeError(f"lp_conv: input {tuple(x.shape)} {x.dtype} does not hold {B}x{H}x{W} rows of {Cin} channels")
    taps = 4 if deconv else ksize * ksize
    if wq.dtype != _TORCH[dtype] or wq.numel() != Cout * taps * Cin:
        raise RuntimeError("lp_conv: weight size / type mismatch")
    _chk(alpha, torch.float32, "alpha"); _chk(beta, torch.float32, "beta")
    if alpha.numel() != Cout or beta.numel() != Cout:
        raise RuntimeError("lp_conv: alpha / beta must have Cout elements")
    rows_out = B * H * W * (4 if deconv else 1)
    if out is None:
        out = _narrow_empty(rows_out, Cout, out_dtype, x.device)
    if out.shape[0] != rows_out or out.dtype != _TORCH[out_dtype] or not out.is_contiguous() or out.shape[1] < out_col0 + Cout:
        raise RuntimeError("lp_conv: destination does not match the output map")
    check(native.lib().rd_lp_conv(_p(x), dtype, B, H, W, Cin, x.shape[1], _p(wq), 2 if deconv else ksize, int(deconv), _p(alpha), _p(beta),
                                  int(relu), _p(out), out_dtype, Cout, out.shape[1], int(out_col0), _stream()), "rd_lp_conv")
    return out


def _kernel_layout(conv):
    """nn.Conv2d [Cout, Cin, kh, kw] -> [Cout, taps, Cin]; nn.ConvTranspose2d [Cin, Cout, kh, kw] -> [Cout, taps, Cin]."""
    w = conv.weight.detach().float()
    if isinstance(conv, nn.ConvTranspose2d):
        return w.permute(1, 2, 3, 0).reshape(w.shape[1], -1, w.shape[0]).contiguous()
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1, w.shape[1]).contiguous()


def _fold(bn):
    rstd = torch.rsqrt(bn.running_var.detach().float() + bn.eps)
    scale = bn.weight.detach().float() * rstd
    return scale, bn.bias.detach().float() - bn.running_mean.detach().float() * scale


class LowpDenseEnc:
    """BaseBEVBackboneV2.dense_enc in bf16 / fp8 storage.  Layer list (as the reference module): blocks[1] (6 convs) on x_conv5,
    deblocks[0] (ConvTranspose2d k2 s2), blocks[0] (6 convs, the first on cat(x_conv4, up))."""

    def __init__(self, module, dtype):
        if dtype not in (BF16, FP8):
            raise ValueError("dtype must be lowp.BF16 or lowp.FP8")
        if module.training:
            raise RuntimeError("LowpDenseEnc folds eval-mode BatchNorm: call module.eval() first")
        self.dtype = dtype
        self.layers = {}                       # name -> dict(wq, w_scale, bn_scale, bn_shift, Cin, Cout, ksize, deconv)
        self.order = []
        for bi in (1, 0):
            mods = [m for m in module.blocks[bi].children() if not isinstance(m, (nn.ZeroPad2d, nn.ReLU))]
            for li in range(0, len(mods), 2):
                self._add(f"blocks.{bi}.{li // 2}", mods[li], mods[li + 1])
            if bi == 1:
                de = module.deblocks[0]
                self._add("deblocks.0", de[0], de[1])
        self.act_scale = {}                    # tensor name -> activation scale (fp8); bf16: all 1
        self._alpha_beta = {}

    def _add(self, name, conv, bn):
        deconv = isinstance(conv, nn.ConvTranspose2d)
        if deconv and (conv.kernel_size != (2, 2) or conv.stride != (2, 2)):
            raise NotImplementedError("LowpDenseEnc: the DenseEnc up-sampling is ConvTranspose2d(k2, s2)")
        if not deconv and (conv.kernel_size != (3, 3) or conv.stride != (1, 1)):
            raise NotImplementedError("LowpDenseEnc: DenseEnc convolutions are 3x3 stride 1")
        wq, ws = lp_quant_weights(_kernel_layout(conv), self.dtype)
        sc, sh = _fold(bn)
        self.layers[name] = dict(wq=wq, w_scale=ws, bn_scale=sc.contiguous(), bn_shift=sh.contiguous(), Cin=conv.in_channels,
                                 Cout=conv.out_channels, ksize=conv.kernel_size[0], deconv=deconv, conv=conv, bn=bn)
        self.order.append(name)

    # ---- calibration (fp8): per-tensor activation scales from one fp32 pass over the same layers
    def calibrate(self, x_conv4, x_conv5):
        """x_conv4 (B, 256, H, W) / x_conv5 (B, 256, H/2, W/2): fp32 CUDA maps of a representative batch.  Runs every layer on the
        fp32 kernels (conv + folded BatchNorm + ReLU) and records max|activation| per tensor."""
        from . import autograd as A
        from . import dense as D
        from .pcdet.models.backbones_2d.base_bev_backbone import _PaddedView
        amax = {}

        def note(name, t):
            amax[name] = float(lp_amax(t.contiguous().view(-1)))

        def run(name, state):
            L = self.layers[name]
            conv = L["conv"]
            if not L["deconv"] and conv.padding[0] == 0:
                conv = _PaddedView(conv, 1)              # ZeroPad2d(1) + Conv2d(p = 0) of the reference Sequential
            return D.conv_bn_act(None, conv, L["bn"], None, act=1, return_rows=True, in_rows=state)

        with torch.no_grad():
            r4, B, H4, W4 = A.nchw_to_rows(x_conv4)
            r5, _, H5, W5 = A.nchw_to_rows(x_conv5)
            note("in5", r5); note("in4", r4)
            state = (r5.contiguous(), B, H5, W5)
            for n in [n for n in self.order if n.startswith("blocks.1.")]:
                state = run(n, state)
                note(n, state[0])
            state = run("deblocks.0", state)
            note("deblocks.0", state[0])
            state = (torch.cat((r4, state[0]), dim=1).contiguous(), B, H4, W4)
            for n in [n for n in self.order if n.startswith("blocks.0.")]:
                state = run(n, state)
                note(n, state[0])
        if self.dtype == FP8:
            self.act_scale = {k: (v / FP8_MAX if v > 0 else 1.0) for k, v in amax.items()}
            # the concat shares ONE scale (both halves feed the same convolution): the larger of the two
            s = max(self.act_scale["in4"], self.act_scale["deblocks.0"])
            self.act_scale["in4"] = self.act_scale["deblocks.0"] = s
        self._alpha_beta = {}
        return amax

    def _scale(self, name):
        return self.act_scale.get(name, 1.0) if self.dtype == FP8 else 1.0

    def _ab(self, name, s_in, s_out):
        """alpha = s_in * w_scale * bn_scale / s_out, beta = bn_shift / s_out."""
        key = (name, s_in, s_out)
        hit = self._alpha_beta.get(key)
        if hit is None:
            L = self.layers[name]
            hit = self._alpha_beta[key] = ((L["w_scale"] * L["bn_scale"] * (s_in / s_out)).contiguous(), (L["bn_shift"] / s_out).contiguous())
        return hit

    def forward_rows(self, x4_rows, x5_rows, B, H4, W4, out_dtype=F32):
        """x4_rows (B*H4*W4, 256), x5_rows (B*H4/2*W4/2, 256) fp32 -> (up, feat) rows (B*H4*W4, 256) in out_dtype."""
        dt = self.dtype
        if self.dtype == FP8 and not self.act_scale:
            raise RuntimeError("LowpDenseEnc(fp8): call calibrate() first")
        H5, W5 = H4 // 2, W4 // 2
        dev = x4_rows.device
        x = lp_cast(x5_rows, dt, 1.0 / self._scale("in5"))
        s_in = self._scale("in5")
        n1 = [n for n in self.order if n.startswith("blocks.1.")]
        for n in n1:
            L = self.layers[n]
            s_out = self._scale(n)
            a, b = self._ab(n, s_in, s_out)
            x = lp_conv(x, dt, B, H5, W5, L["Cin"], L["wq"], 3, a, b, True, dt, L["Cout"])
            s_in = s_out
        # up-sampling writes its narrow output straight into the right half of the concat buffer; x_conv4 is cast into the left half
        L = self.layers["deblocks.0"]
        Cup = L["Cout"]
        s_cat = self._scale("deblocks.0")
        cat = _narrow_empty(B * H4 * W4, x4_rows.shape[1] + Cup, dt, dev)
        a, b = self._ab("deblocks.0", s_in, s_cat)
        lp_conv(x, dt, B, H5, W5, L["Cin"], L["wq"], 2, a, b, True, dt, L["Cout"], out=cat, out_col0=x4_rows.shape[1], deconv=True)
        lp_cast(x4_rows, dt, 1.0 / s_cat, out=cat, out_col0=0)
        # (`up` is also a module output, spatial_features_2d_8x: taken from the concat buffer's right half below)
        n0 = [n for n in self.order if n.startswith("blocks.0.")]
        x, s_in = cat, s_cat
        for k, n in enumerate(n0):
            L = self.layers[n]
            last = k == len(n0) - 1
            s_out = 1.0 if (last and out_dtype == F32) else self._scale(n)
            a, b = self._ab(n, s_in, s_out)
            x = lp_conv(x, dt, B, H4, W4, L["Cin"], L["wq"], 3, a, b, True, F32 if (last and out_dtype == F32) else dt, L["Cout"])
            s_in = s_out
        feat = x
        up = cat[:, x4_rows.shape[1]:].contiguous()
        if out_dtype == F32:
            up = lp_uncast(up, dt, s_cat)
        return up, feat

    def forward(self, x_conv4, x_conv5):
        """(B, 256, H, W) fp32 maps -> (spatial_features_2d_8x, spatial_features_2d) fp32 (channels-last memory)."""
        from . import autograd as A
        r4, B, H4, W4 = A.nchw_to_rows(x_conv4)
        r5, _, _, _ = A.nchw_to_rows(x_conv5)
        up, feat = self.forward_rows(r4.contiguous(), r5.contiguous(), B, H4, W4)
        return A.rows_to_nchw(up, B, H4, W4), A.rows_to_nchw(feat, B, H4, W4)
