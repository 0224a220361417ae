"""DCNv2 module + autograd function on the MI355X kernels.

Interface of the reference's pcdet/ops/basicblock/modules/modulated_deform_conv.py:14-64 (ModulatedDeformConv) and
functions/modulated_deform_conv_func.py:15-56 (ModulatedDeformConvFunction), deformable_groups = groups = 1.
Quirk kept on purpose: `bias=False` does NOT remove the bias -- the parameter exists, is initialised U(-1/sqrt(fan_in), ..),
is merely frozen, and the forward always adds it (modulated_deform_conv.py:36-49, modulated_deform_conv_cuda.cu:112).
"""
import math

import torch
import torch.nn as nn
from torch.nn import init
from torch.nn.modules.utils import _pair

import os

from radardistill_amd import autograd as A
from radardistill_amd import kernels as K

# DCNv2 as explicit deformed columns + plain GEMMs (default) or with the sampling fused into the GEMM's operand staging (RD_DCN_COLS=0)
DCN_COLUMNS = os.environ.get("RD_DCN_COLS", "1") != "0"


class _DCNFn(torch.autograd.Function):
    """x_rows (B*H*W, Cin), om_rows (B*Ho*Wo, S): offsets in columns [0, 2*taps), mask in [2*taps, 3*taps)
    (pre-sigmoid when sig=True) -> out rows (B*Ho*Wo, Cout)."""

    @staticmethod
    def forward(ctx, x_rows, om_rows, weight, bias, geom, sig):
        B, H, W, k, stride, pad = geom
        taps = k * k
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        Cout, Cin = weight.shape[0], weight.shape[1]
        S = om_rows.shape[1]
        if om_rows.shape != (B * Ho * Wo, S) or S < 3 * taps or x_rows.shape != (B * H * W, Cin):
            raise RuntimeError("DCN: offset/mask/input shapes do not match the geometry")
        mask_base = om_rows[0:1, 2 * taps:]           # view: data_ptr of the first mask channel
        samp_idx, samp_w = K.dcn_prep(om_rows, S, mask_base, S, sig, B, H, W, Ho, Wo, k, stride, pad)
        rows_o = B * Ho * Wo
        col = None
        A.note_param_use(weight, bias)
        if DCN_COLUMNS and x_rows.is_cuda and Cin % 4 == 0:
            # column form: the sampled, modulated input rows are written once (rows x taps*Cin) and the convolution is a plain GEMM
            # over them; the backward's weight gradient re-uses the same columns (kept: 75 MB at the CMA shapes)
            col = K.dcn_columns(x_rows, samp_idx, samp_w)
            lin = A.linear_spec(rows_o).fwd_ix
            if A._b3_wsplit(taps * Cin, Cout):
                # (the fragment-major image of [Cout][taps][Cin] read as a 1-tap GEMM with K = taps * Cin is the same bytes)
                frag = Cin % 16 == 0 and K.wants_frag_weights(lin, rows_o, rows_o, taps * Cin, Cout, 1)
                out = K.conv_fwd(col, A.operand_weight_split(weight, Cout, Cin, taps, 1, frag=frag), 1, bias.detach(), rows_o, Cout, lin,
                                 w_split=2 if frag else True)
            else:
                out = K.conv_fwd(col, A.kernel_weight(weight, Cout, Cin, taps, 1), 1, bias.detach(), rows_o, Cout, lin)
        else:         # sampling fused into the GEMM's operand staging (index mode 3)
            ix = K.conv_index_deform(samp_idx, samp_w)
            wk = A.kernel_weight(weight, Cout, Cin, taps, 1)
            out = K.conv_fwd(x_rows, wk, taps, bias.detach(), rows_o, Cout, ix)
        ctx.geom, ctx.sig, ctx.out_hw = geom, sig, (Ho, Wo)
        ctx.has_col = col is not None
        if col is not None:
            ctx.save_for_backward(x_rows, om_rows, weight, col)
        else:
            ctx.save_for_backward(x_rows, om_rows, weight, samp_idx, samp_w)
        ctx.bias_grad = bias.requires_grad
        ctx.bias_ref = bias
        return out

    @staticmethod
    def backward(ctx, go):
        if ctx.has_col:
            x_rows, om_rows, weight, col = ctx.saved_tensors
        else:
            x_rows, om_rows, weight, samp_idx, samp_w = ctx.saved_tensors
        B, H, W, k, stride, pad = ctx.geom
        Ho, Wo = ctx.out_hw
        taps = k * k
        Cout, Cin = weight.shape[0], weight.shape[1]
        S = om_rows.shape[1]
        go = go.contiguous()
        rows_o = B * Ho * Wo
        wk = A.kernel_weight(weight, Cout, Cin, taps, 1)
        # column gradient colgrad[j][t][c] = sum_n go[j][n] W[n][t][c]: a linear layer with taps*Cin outputs
        w6 = K.weight_layout(wk, Cout, Cin, taps, 6, False)
        colgrad = K.conv_fwd(go, w6, 1, None, rows_o, taps * Cin, A.linear_spec(rows_o).fwd_ix)
        g_om = torch.zeros_like(om_rows) if S > 3 * taps else torch.empty_like(om_rows)
        gx = K.dcn_bwd_data(x_rows, colgrad, om_rows, S, om_rows[0:1, 2 * taps:], S, ctx.sig, B, H, W, Ho, Wo, k, stride, pad,
                            g_om, S, g_om[0:1, 2 * taps:], S)
        # parameter gradients: nobody reads them before the optimizer -> side stream (autograd.param_grad_stream)
        if ctx.has_col:
            lin = A.linear_spec(rows_o).fwd_ix
            gw = A.param_grad_stream(lambda: K.weight_layout(K.conv_wgrad(col, go, 1, lin), Cout, Cin, taps, 4, False, out_shape=tuple(weight.shape)),
                                     col, go, param=weight)
        else:
            ix = K.conv_index_deform(samp_idx, samp_w)
            gw = A.param_grad_stream(lambda: K.weight_layout(K.conv_wgrad(x_rows, go, taps, ix), Cout, Cin, taps, 4, False,
                                                             out_shape=tuple(weight.shape)), x_rows, go, samp_idx, samp_w, param=weight)
        gb = A.param_grad_stream(lambda: K.colsum(go) if Cout % 4 == 0 else go.sum(0), go, param=ctx.bias_ref) if ctx.bias_grad else None
        return gx, g_om, gw, gb, None, None


class ModulatedDeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, groups=1, deformable_groups=1,
                 im2col_step=64, bias=True):
        super().__init__()
        if groups != 1 or deformable_groups != 1 or dilation != 1:
            raise NotImplementedError("RadarDistill uses groups = deformable_groups = dilation = 1")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups, self.im2col_step, self.use_bias = groups, deformable_groups, im2col_step, bias
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        self.bias = nn.Parameter(torch.Tensor(out_channels))
        self.reset_parameters()
        if not self.use_bias:
            self.bias.requires_grad = False

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
        bound = 1 / math.sqrt(fan_in)
        init.uniform_(self.bias, -bound, bound)

    def forward_rows(self, x_rows, B, H, W, om_rows, sig):
        """Fast path used by ConvNeXtBlock: om_rows is the raw 27-channel conv_offset_mask output (sigmoid fused)."""
        geom = (B, H, W, self.kernel_size[0], self.stride[0], self.padding[0])
        out = _DCNFn.apply(x_rows, om_rows, self.weight, self.bias, geom, sig)
        k, s, p = self.kernel_size[0], self.stride[0], self.padding[0]
        return out, (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1

    def forward(self, input, offset, mask):
        """Reference signature: input (B,C,H,W), offset (B,2*k*k,Ho,Wo), mask (B,k*k,Ho,Wo) (already sigmoid-ed)."""
        taps = self.kernel_size[0] * self.kernel_size[1]
        assert 2 * self.deformable_groups * taps == offset.shape[1]
        assert self.deformable_groups * taps == mask.shape[1]
        x_rows, B, H, W = A.nchw_to_rows(input)
        om = torch.cat((offset, mask), dim=1).permute(0, 2, 3, 1).reshape(-1, 3 * taps).contiguous()
        out, Ho, Wo = self.forward_rows(x_rows, B, H, W, om, False)
        return A.rows_to_nchw(out, B, Ho, Wo)
