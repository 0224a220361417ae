"""SparseEnc: the PillarRes18 2-D sparse backbone on the MI355X kernels.

Module tree, constructor signature, batch_dict keys and state_dict names follow the reference's
pcdet/models/backbones_3d/spconv_backbone_2d.py:9-112,208-324; the arithmetic is re-designed:
  * frozen / eval:  conv + folded BatchNorm + (residual) + ReLU is ONE implicit-GEMM kernel per layer,
  * training:       conv kernel (BatchNorm statistics accumulated in its epilogue) -> one fused
                    normalise(+residual)+ReLU pass; backward = data-gradient conv, weight-gradient GEMM, fused BN backward,
  * x_conv4.dense() and the dense conv5 stage stay channels-last (rows x channels), so conv5 runs the same kernel.
"""
from functools import partial

import torch
import torch.nn as nn

from radardistill_amd import autograd as A
from radardistill_amd import dense as D
from radardistill_amd import kernels as K
from radardistill_amd.sparse import SparseConvTensor
from radardistill_amd.pcdet.utils.spconv_utils import replace_feature, spconv


_frozen = D.frozen


class _SparseConvBNReLU(spconv.SparseSequential):
    """post_act_block: children '0' (sparse conv), '1' (BatchNorm1d), '2' (ReLU) -- spconv_backbone_2d.py:9-28."""

    def forward(self, x):
        conv, bn = self[0], self[1]
        spec, lvl = conv._spec_and_level(x)
        if not bn.training and _frozen(conv, bn):
            scale, shift = A.bn_eval_scale_shift(bn)
            feats = A.conv_inference(x.features, conv.weight, conv.bias, spec, conv.out_channels, scale, shift, None, True)
        elif bn.training:
            feats = A.conv_bn_act_train(x.features, conv.weight, conv.bias, spec, conv.out_channels, bn, None, 1)
        else:
            raw = A.conv(x.features, conv.weight, conv.bias, spec, conv.out_channels, None)
            feats = A.bn_act_eval(raw, bn, None, act=1)
        out = SparseConvTensor(feats, lvl.coords, [lvl.H, lvl.W], x.batch_size, _level=lvl)
        return out


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type='subm', norm_fn=None):
    if conv_type == 'subm':
        conv = spconv.SubMConv2d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == 'spconv':
        conv = spconv.SparseConv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False, indice_key=indice_key)
    else:
        raise NotImplementedError
    return _SparseConvBNReLU(conv, norm_fn(out_channels), nn.ReLU())


class _DenseConvBNReLU(nn.Sequential):
    """post_act_block_dense: children '0' Conv2d, '1' BatchNorm2d, '2' ReLU (spconv_backbone_2d.py:31-38), channels-last."""

    def forward(self, x):
        conv, bn = self[0], self[1]
        return dense_conv_bn_act(x, conv, bn, None, act=1)


def dense_conv_bn_act(x, conv, bn, residual_rows=None, act=1, return_rows=False):
    return D.conv_bn_act(x, conv, bn, residual_rows, act, return_rows)


def post_act_block_dense(in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, norm_fn=None):
    assert dilation == 1
    return _DenseConvBNReLU(
        nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding=padding, dilation=dilation, bias=False),
        norm_fn(out_channels),
        nn.ReLU(),
    )


class SparseBasicBlock(spconv.SparseModule):
    """conv(bias) -> BN -> ReLU -> conv(bias) -> BN -> (+identity) -> ReLU (spconv_backbone_2d.py:41-77)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, norm_fn=None, downsample=None, indice_key=None):
        super().__init__()
        assert norm_fn is not None and downsample is None and stride == 1
        bias = norm_fn is not None
        self.conv1 = spconv.SubMConv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=bias, indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = spconv.SubMConv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=bias, indice_key=indice_key)
        self.bn2 = norm_fn(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        spec = x._level.subm_spec()
        f = x.features
        C = self.conv1.out_channels
        if not self.bn1.training and _frozen(self.conv1, self.bn1, self.conv2, self.bn2):
            s1, h1 = A.bn_eval_scale_shift(self.bn1)
            s2, h2 = A.bn_eval_scale_shift(self.bn2)
            y = A.conv_inference(f, self.conv1.weight, self.conv1.bias, spec, C, s1, h1, None, True)
            y = A.conv_inference(y, self.conv2.weight, self.conv2.bias, spec, C, s2, h2, f, True)
        elif self.bn1.training:
            m = self._modules          # (plain dictionary reads instead of nn.Module.__getattr__: autograd.bn_tensors)
            w1, b1 = A.conv_params(m['conv1'])
            w2, b2 = A.conv_params(m['conv2'])
            y = A.conv_bn_act_train(f, w1, b1, spec, C, m['bn1'], None, 1)
            y = A.conv_bn_act_train(y, w2, b2, spec, C, m['bn2'], f, 1)
        else:
            y = A.conv(f, self.conv1.weight, self.conv1.bias, spec, C, None)
            y = A.bn_act_eval(y, self.bn1, None, act=1)
            y = A.conv(y, self.conv2.weight, self.conv2.bias, spec, C, None)
            y = A.bn_act_eval(y, self.bn2, f, act=1)
        return replace_feature(x, y)


class BasicBlock(nn.Module):
    """Dense residual block of conv5 (spconv_backbone_2d.py:80-112), channels-last rows."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, norm_fn=None, downsample=None):
        super().__init__()
        assert norm_fn is not None and downsample is None and stride == 1
        bias = norm_fn is not None
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=bias)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=bias)
        self.bn2 = norm_fn(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        rows, B, H, W = A.nchw_to_rows(x)
        y = dense_conv_bn_act(x, self.conv1, self.bn1, None, act=1)
        return dense_conv_bn_act(y, self.conv2, self.bn2, rows, act=1)


class PillarRes18BackBone8x(nn.Module):
    IN_PREFIX = ""

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(grid_size[1]), int(grid_size[0])]          # grid_size[[1, 0]]
        block = post_act_block
        dense_block = post_act_block_dense
        self.conv1 = spconv.SparseSequential(
            SparseBasicBlock(32, 32, norm_fn=norm_fn, indice_key='res1'),
            SparseBasicBlock(32, 32, norm_fn=norm_fn, indice_key='res1'),
        )
        self.conv2 = spconv.SparseSequential(
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
            SparseBasicBlock(64, 64, norm_fn=norm_fn, indice_key='res2'),
            SparseBasicBlock(64, 64, norm_fn=norm_fn, indice_key='res2'),
        )
        self.conv3 = spconv.SparseSequential(
            block(64, 128, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
            SparseBasicBlock(128, 128, norm_fn=norm_fn, indice_key='res3'),
            SparseBasicBlock(128, 128, norm_fn=norm_fn, indice_key='res3'),
        )
        self.conv4 = spconv.SparseSequential(
            block(128, 256, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv4', conv_type='spconv'),
            SparseBasicBlock(256, 256, norm_fn=norm_fn, indice_key='res4'),
            SparseBasicBlock(256, 256, norm_fn=norm_fn, indice_key='res4'),
        )
        norm_fn = partial(nn.BatchNorm2d, eps=1e-3, momentum=0.01)
        self.conv5 = nn.Sequential(
            dense_block(256, 256, 3, norm_fn=norm_fn, stride=2, padding=1),
            BasicBlock(256, 256, norm_fn=norm_fn),
            BasicBlock(256, 256, norm_fn=norm_fn),
        )
        self.num_point_features = 256
        self.backbone_channels = {'x_conv1': 32, 'x_conv2': 64, 'x_conv3': 128, 'x_conv4': 256, 'x_conv5': 256}

    def prepare(self, batch_dict):
        """Build the input SparseConvTensor and the whole 4-level active-site pyramid (rank grids + neighbour tables) before any
        convolution is enqueued (3 small device->host count reads per branch)."""
        p = self.IN_PREFIX
        if p + 'pillar_features' not in batch_dict:
            return
        x = SparseConvTensor(features=batch_dict[p + 'pillar_features'], indices=batch_dict[p + 'pillar_coords'].int(),
                             spatial_shape=self.sparse_shape, batch_size=batch_dict['batch_size'])
        lvl = x._level
        lvl.subm_spec()
        for _ in range(3):
            lvl, _ = lvl.down()
            lvl.subm_spec()
        batch_dict[p + '_sparse_input'] = x

    def forward_sparse(self, batch_dict):
        """The data-dependent half: four sparse stages and `x_conv4.dense()` -> the (B, 256, H/8, W/8) map.  Everything after it
        (conv5 and the 2-D modules) has static shapes; radardistill_amd/graphs.py replays that part as one HIP graph."""
        p = self.IN_PREFIX
        x = batch_dict.pop(p + '_sparse_input', None)
        if x is None:
            self.prepare(batch_dict)
            x = batch_dict.pop(p + '_sparse_input')
        x_conv1 = self.conv1(x)
        x_conv2 = self.conv2(x_conv1)
        x_conv3 = self.conv3(x_conv2)
        x_conv4 = self.conv4(x_conv3).dense()
        batch_dict[p + '_sparse_stages'] = (x_conv1, x_conv2, x_conv3)
        return x_conv4

    def forward(self, batch_dict):
        p = self.IN_PREFIX
        x_conv4 = self.forward_sparse(batch_dict)
        x_conv1, x_conv2, x_conv3 = batch_dict.pop(p + '_sparse_stages')
        x_conv5 = self.conv5(x_conv4)
        batch_dict.update({p + 'multi_scale_2d_features': {
            'x_conv1': x_conv1, 'x_conv2': x_conv2, 'x_conv3': x_conv3, 'x_conv4': x_conv4, 'x_conv5': x_conv5}})
        batch_dict.update({p + 'multi_scale_2d_strides': {
            'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8, 'x_conv5': 16}})
        return batch_dict
