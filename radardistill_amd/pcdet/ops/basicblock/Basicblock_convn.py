"""ConvNeXt-V2 block with optional DCNv2 down-sampling (CMA building block), channels-last end to end.

Module / parameter names follow the reference's pcdet/ops/basicblock/modules/Basicblock_convn.py:10-95.  The reference
permutes NCHW -> NHWC -> NCHW around the MLP (two copies per block); here maps are channels-last throughout, so the
permutes are views.  conv_offset_mask1, the DCN, pwconv1 and pwconv2 run on the implicit-GEMM MFMA kernel;
chunk/cat/sigmoid of the offset-mask tensor are fused into the DCN sampling-table kernel.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from radardistill_amd import autograd as A
from radardistill_amd import dense as D
from .modulated_deform_conv import ModulatedDeformConv


class LayerNorm(nn.Module):
    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.data_format = data_format
        if self.data_format not in ["channels_last", "channels_first"]:
            raise NotImplementedError
        self.normalized_shape = (normalized_shape,)

    def forward(self, x):
        if self.data_format == "channels_last":
            C = self.normalized_shape[0]
            if x.is_cuda and x.dtype == torch.float32 and C % 4 == 0 and C <= 1024 and x.is_contiguous():
                return A.layer_norm_rows(x.reshape(-1, C), self.weight, self.bias, self.eps).view(x.shape)      # layernorm.hip
            return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        x = (x - u) / torch.sqrt(s + self.eps)
        return self.weight[:, None, None] * x + self.bias[:, None, None]


class GRN(nn.Module):
    """Global Response Normalization over (H, W) of a (B, H, W, C) tensor."""

    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.zeros(1, 1, 1, dim))
        self.beta = nn.Parameter(torch.zeros(1, 1, 1, dim))

    def forward(self, x):
        Gx = torch.norm(x, p=2, dim=(1, 2), keepdim=True)
        Nx = Gx / (Gx.mean(dim=-1, keepdim=True) + 1e-6)
        return self.gamma * (x * Nx) + self.beta + x


class ConvNeXtBlock(nn.Module):
    def __init__(self, dim, downsample=False, deformable_groups=1):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.grn = GRN(4 * dim)
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.downsample = downsample
        if self.downsample:
            offset_mask_channels = 3 * 3 * (2 + 1)
            # NOT zero-initialised (unlike Basicblock.init_offset): offsets are random from step 0, as in the reference
            self.conv_offset_mask1 = nn.Conv2d(dim, deformable_groups * offset_mask_channels, kernel_size=3, stride=2, padding=1, bias=True)
            self.down_layer = ModulatedDeformConv(dim, dim, stride=2, kernel_size=3, padding=1, deformable_groups=deformable_groups, bias=False)

    def forward(self, x):
        rows, B, H, W = A.nchw_to_rows(x)
        if self.downsample:
            om, _, Ho, Wo = D.conv_bn_act(None, self.conv_offset_mask1, None, None, act=0, return_rows=True, in_rows=(rows, B, H, W))
            # offset = cat(o1, o2) keeps channels 0..17, mask = sigmoid(channels 18..26) (Basicblock_convn.py:40-43)
            rows, H, W = self.down_layer.forward_rows(rows, B, H, W, om, True)
        C = rows.shape[1]
        identity = rows
        y = A.dwconv(rows, self.dwconv, B, H, W)                                             # depthwise 7x7 (dwconv.hip)
        norm = self.norm
        if y.is_cuda and y.dtype == torch.float32 and norm.data_format == "channels_last" and C % 4 == 0 and C <= 1024 and y.is_contiguous():
            y = A.layer_norm_rows(y, norm.weight, norm.bias, norm.eps)                       # layernorm.hip, rows in, rows out (no views)
        else:
            y = norm(y.view(B, H, W, C)).reshape(-1, C)
        y = D.linear_rows(y, self.pwconv1)
        if y.is_cuda and isinstance(self.act, nn.GELU) and getattr(self.act, 'approximate', 'none') == 'none':
            y = A.gelu_grn(y, self.grn, B)                                                   # GELU + GRN fused (convnext.hip)
        else:
            y = self.grn(self.act(y).view(B, H, W, 4 * C)).reshape(-1, 4 * C)
        if y.is_cuda:
            return A.rows_to_nchw(D.linear_rows(y, self.pwconv2, residual=identity), B, H, W)          # + identity in the GEMM's epilogue
        return A.rows_to_nchw(D.linear_rows(y.reshape(-1, 4 * C), self.pwconv2) + identity, B, H, W)
