"""DenseEnc: BaseBEVBackboneV2 (pcdet/models/backbones_2d/base_bev_backbone.py:205-308) on the MFMA conv kernel.

nn.Sequential containers keep the reference's child indices (ZeroPad2d 0, Conv2d 1, BN 2, ReLU 3, Conv 4, ...), so state
dict keys are `blocks.{0,1}.{1,4,..}.weight`, `deblocks.0.0.weight` ...  ZeroPad2d(1)+Conv(p=0) is executed as one
zero-padded conv; the frozen teacher runs conv+folded-BN+ReLU as a single kernel per layer.
"""
import numpy as np
import torch
import torch.nn as nn

from radardistill_amd import autograd as A
from radardistill_amd import dense as D


def run_block(seq, x):
    """Execute a blocks[i] Sequential: [ZeroPad2d, Conv, BN, ReLU, (Conv, BN, ReLU)*]."""
    mods = list(seq.children())
    i = 0
    pad_next = 0
    rows_state = A.nchw_to_rows(x)
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.ZeroPad2d):
            pad_next = int(m.padding[0])
            i += 1
            continue
        assert isinstance(m, nn.Conv2d) and isinstance(mods[i + 1], (nn.BatchNorm2d, nn.SyncBatchNorm)) and isinstance(mods[i + 2], nn.ReLU)
        conv = m
        if pad_next:
            conv = _PaddedView(m, pad_next)
            pad_next = 0
        out, B, Ho, Wo = D.conv_bn_act(None, conv, mods[i + 1], None, act=1, return_rows=True, in_rows=rows_state)
        rows_state = (out, B, Ho, Wo)
        i += 3
    return A.rows_to_nchw(*rows_state)


class _PaddedView:
    """nn.Conv2d(padding=0) preceded by ZeroPad2d(p): same parameters, geometry with padding p."""

    def __init__(self, conv, pad):
        self._c = conv
        self.padding = (pad, pad)

    def __getattr__(self, k):
        return getattr(self._c, k)

    def parameters(self):
        return self._c.parameters()


class BaseBEVBackboneV2(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        layer_nums = self.model_cfg.LAYER_NUMS
        num_filters = self.model_cfg.NUM_FILTERS
        assert len(layer_nums) == len(num_filters) == 2
        num_upsample_filters = self.model_cfg.NUM_UPSAMPLE_FILTERS
        upsample_strides = self.model_cfg.UPSAMPLE_STRIDES
        assert len(num_upsample_filters) == len(upsample_strides)
        num_levels = len(layer_nums)
        self.blocks = nn.ModuleList()
        self.deblocks = nn.ModuleList()
        for idx in range(num_levels):
            cin = num_filters[idx] * 2 if idx == 0 else num_filters[idx]
            cur_layers = [nn.ZeroPad2d(1), nn.Conv2d(cin, num_filters[idx], kernel_size=3, stride=1, padding=0, bias=False),
                          nn.BatchNorm2d(num_filters[idx], eps=1e-3, momentum=0.01), nn.ReLU()]
            for k in range(layer_nums[idx]):
                cur_layers.extend([nn.Conv2d(num_filters[idx], num_filters[idx], kernel_size=3, padding=1, bias=False),
                                   nn.BatchNorm2d(num_filters[idx], eps=1e-3, momentum=0.01), nn.ReLU()])
            self.blocks.append(nn.Sequential(*cur_layers))
            if len(upsample_strides) > 0:
                stride = upsample_strides[idx]
                if stride >= 1:
                    self.deblocks.append(nn.Sequential(
                        nn.ConvTranspose2d(num_filters[idx], num_upsample_filters[idx] * 2, upsample_strides[idx],
                                           stride=upsample_strides[idx], bias=False),
                        nn.BatchNorm2d(num_upsample_filters[idx] * 2, eps=1e-3, momentum=0.01), nn.ReLU()))
                else:
                    stride = int(np.round(1 / stride))
                    self.deblocks.append(nn.Sequential(
                        nn.Conv2d(num_filters[idx], num_upsample_filters[idx], stride, stride=stride, bias=False),
                        nn.BatchNorm2d(num_upsample_filters[idx], eps=1e-3, momentum=0.01), nn.ReLU()))
        c_in = sum(num_upsample_filters)
        if len(upsample_strides) > num_levels:
            self.deblocks.append(nn.Sequential(
                nn.ConvTranspose2d(c_in, c_in, upsample_strides[-1], stride=upsample_strides[-1], bias=False),
                nn.BatchNorm2d(c_in, eps=1e-3, momentum=0.01), nn.ReLU()))
        self.num_bev_features = c_in
        self.deblocks = self.deblocks[1:]          # base_bev_backbone.py:282

    def dense_enc(self, x_conv4, x_conv5):
        x = run_block(self.blocks[1], x_conv5)
        de = self.deblocks[0]
        up = D.conv_bn_act(x, de[0], de[1], None, act=1)
        feat = run_block(self.blocks[0], A.cat_channels(x_conv4, up))
        return up, feat

    def _lowp_engine(self, x_conv4, x_conv5):
        """MODEL.BACKBONE_2D.PRECISION: bf16 | fp8 (BASELINE configs[2] / [4]; default fp32): the FROZEN eval-mode DenseEnc on the
        low-precision kernels (radardistill_amd/lowp.py).  Built once per weight version; fp8 scales are calibrated on the first
        batch seen (a deployment would ship them with the checkpoint)."""
        from radardistill_amd import lowp as LP
        prec = str(self.model_cfg.get('PRECISION', 'fp32')).lower()
        if prec in ('fp32', 'f32'):
            return None
        if prec not in ('bf16', 'fp8'):
            raise ValueError(f"BACKBONE_2D.PRECISION: {prec!r} (fp32, bf16 or fp8)")
        if self.training or any(p.requires_grad for p in self.parameters()) or not x_conv4.is_cuda:
            raise RuntimeError("BACKBONE_2D.PRECISION bf16 / fp8 is an inference mode of the frozen teacher DenseEnc (eval mode, no gradients)")
        ver = tuple(p._version for p in self.parameters()) + tuple(b._version for b in self.buffers())
        eng = getattr(self, '_lowp', None)
        if eng is None or eng[0] != (prec, ver):
            e = LP.LowpDenseEnc(self, LP.BF16 if prec == 'bf16' else LP.FP8)
            if prec == 'fp8':
                e.calibrate(x_conv4, x_conv5)
            eng = self._lowp = ((prec, ver), e)
        return eng[1]

    def forward(self, data_dict):
        sf = data_dict['multi_scale_2d_features']
        eng = self._lowp_engine(sf['x_conv4'], sf['x_conv5'])
        if eng is not None:
            up, feat = eng.forward(sf['x_conv4'], sf['x_conv5'])
        else:
            up, feat = self.dense_enc(sf['x_conv4'], sf['x_conv5'])
        data_dict['spatial_features_2d_8x'] = up
        data_dict['spatial_features_2d'] = feat
        return data_dict
