"""Dense (channels-last) layer helpers shared by the host modules: nn.Conv2d / nn.ConvTranspose2d / nn.Linear /
nn.BatchNorm2d objects are used as PARAMETER CONTAINERS (so state_dict names match the reference); the arithmetic runs
on the implicit-GEMM MFMA kernel and the fused BatchNorm kernels."""
import torch
import torch.nn as nn

from . import autograd as A


def frozen(*mods):
    """True when the fused inference path applies: no gradient wanted through these modules."""
    return (not torch.is_grad_enabled()) or all(not any(p.requires_grad for p in m.parameters()) for m in mods)


def conv_rows(rows, B, H, W, conv, stats=None):
    """rows (B*H*W, Cin) -> (rows_out, Cout), (Ho, Wo).  conv: nn.Conv2d or nn.ConvTranspose2d container (zero padding)."""
    kh, kw = conv.kernel_size
    transposed = isinstance(conv, nn.ConvTranspose2d)
    if conv.dilation != (1, 1) or conv.groups != 1 or conv.stride[0] != conv.stride[1] or conv.padding[0] != conv.padding[1]:
        raise NotImplementedError("implicit-GEMM conv: square stride/padding, no dilation/groups")
    spec = A.dense_conv_spec(B, H, W, kh, kw, conv.stride[0], conv.padding[0], transposed=transposed)
    return spec, conv.out_channels


def conv_bn_act(x, conv, bn=None, residual_rows=None, act=1, return_rows=False, in_rows=None):
    """(B,Cin,H,W) [or in_rows=(rows,B,H,W)] -> conv -> BatchNorm -> (+residual) -> act (0 none, 1 ReLU, 2 GELU)."""
    rows, B, H, W = in_rows if in_rows is not None else A.nchw_to_rows(x)
    spec, Cout = conv_rows(rows, B, H, W, conv)
    Ho, Wo = spec.out_hw
    if bn is None:
        if frozen(conv) and act in (0, 1):
            out = A.conv_inference(rows, conv.weight, conv.bias, spec, Cout, None, None, residual_rows, act == 1)
        else:
            out = A.conv(rows, conv.weight, conv.bias, spec, Cout, None)
            if residual_rows is not None:
                out = out + residual_rows
            if act == 1:
                out = torch.relu(out)
            elif act == 2:
                out = torch.nn.functional.gelu(out)
    elif not bn.training and frozen(conv, bn) and act in (0, 1):
        scale, shift = A.bn_eval_scale_shift(bn)
        out = A.conv_inference(rows, conv.weight, conv.bias, spec, Cout, scale, shift, residual_rows, act == 1)
    elif bn.training:
        w, b = A.conv_params(conv)
        out = A.conv_bn_act_train(rows, w, b, spec, Cout, bn, residual_rows, act)
    else:
        raw = A.conv(rows, conv.weight, conv.bias, spec, Cout, None)
        out = A.bn_act_eval(raw, bn, residual_rows, act=act)
    return (out, B, Ho, Wo) if return_rows else A.rows_to_nchw(out, B, Ho, Wo)


def linear_rows(rows, lin, residual=None):
    """nn.Linear container on (rows, Cin) -> (rows, Cout) through the 1-tap implicit GEMM (+ residual rows in the epilogue)."""
    w, b = A.conv_params(lin)
    return A.conv(rows, w, b, A.linear_spec(rows.shape[0]), lin.out_features, None, residual=residual)
