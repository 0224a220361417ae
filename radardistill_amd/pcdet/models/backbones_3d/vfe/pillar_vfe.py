"""PillarVFE over the padded-voxel input format (pcdet/models/backbones_3d/vfe/pillar_vfe.py:8-123) on the HIP kernels of
hardvox.hip.  Module tree and state_dict names are the reference's (`pfn_layers.0.linear.weight`, `pfn_layers.0.norm.*`).

Two paths, same results:
  * no gradients, one PFN layer with BatchNorm (inference / frozen use): feature assembly + Linear + BN + ReLU + slot max fused in
    ONE kernel (two passes in train mode for the batch statistics) -- rd_pillar_vfe_{stats,max};
  * everything else (training with gradients, several PFN layers, USE_NORM False): rd_pillar_decorate writes the masked slot
    features as rows once, each PFNLayer is the 1-tap implicit GEMM -> BatchNorm rows kernel -> rd_pfn_pool (slot max, and for a
    non-last layer the [x | max] concatenation), every piece an autograd node with a HIP backward.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from radardistill_amd import autograd as A
from radardistill_amd import kernels as K
from .vfe_template import VFETemplate


def _pad32(n):
    return (n + 31) // 32 * 32


class _PfnPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, M, P, last):
        out, argmax = K.pfn_pool_fwd(x.contiguous(), M, P, last)
        ctx.save_for_backward(argmax)
        ctx.dims = (M, P, last)
        return out

    @staticmethod
    def backward(ctx, g):
        (argmax,) = ctx.saved_tensors
        M, P, last = ctx.dims
        return K.pfn_pool_bwd(g.contiguous(), argmax, M, P, last), None, None, None


class PFNLayer(nn.Module):
    """Parameter container of pillar_vfe.py:8-27 (the arithmetic runs fused in rd_pillar_vfe_*)."""

    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        self.last_vfe = last_layer
        self.use_norm = use_norm
        if not self.last_vfe:
            out_channels = out_channels // 2
        if self.use_norm:
            self.linear = nn.Linear(in_channels, out_channels, bias=False)
            self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)
        else:
            self.linear = nn.Linear(in_channels, out_channels, bias=True)
        self.part = 50000          # the reference splits its nn.Linear call above this many rows (pillar_vfe.py:30-36); one launch here
        self.out_channels = out_channels

    def forward_rows(self, x, M, P, cols=None):
        """x (M*P, ld) rows whose logical input columns sit at `cols` (None: the first in_features columns; the rest are zero) ->
        last layer: (M, Cp); else (M*P, 2*Cp) = [x | max], Cp = out_channels rounded up to 32 (padded channels are exact zeros)."""
        Cout, Cin, ld = self.out_channels, self.linear.in_features, x.shape[1]
        Cp = _pad32(Cout)
        w = self.linear.weight
        if Cp != Cout:
            w = F.pad(w, (0, 0, 0, Cp - Cout))
        if cols is not None:
            w = w.new_zeros((Cp, ld)).index_copy(1, cols, w)
        elif ld != Cin:
            w = F.pad(w, (0, ld - Cin))
        b = self.linear.bias
        if b is not None and Cp != Cout:
            b = F.pad(b, (0, Cp - Cout))
        rows = M * P
        if self.use_norm:
            bn = self.norm
            if bn.training:
                if rows <= 1:
                    raise ValueError("Expected more than 1 value per channel when training")
                x = A.conv(x, w, b, A.linear_spec(rows), Cp, None)
                if Cp == Cout:
                    x = A.bn_act_train(x, bn, act=1)
                else:       # padded channels: gamma 1, beta 0, throw-away running statistics
                    rm, rv = F.pad(bn.running_mean, (0, Cp - Cout)), F.pad(bn.running_var, (0, Cp - Cout), value=1.0)
                    x = A.bn_act_train_tensors(x, F.pad(bn.weight, (0, Cp - Cout), value=1.0), F.pad(bn.bias, (0, Cp - Cout)), rm, rv,
                                               float(bn.eps), float(bn.momentum), act=1, modules=(bn,))
                    with torch.no_grad():
                        bn.running_mean.copy_(rm[:Cout]); bn.running_var.copy_(rv[:Cout])
            else:
                x = A.conv(x, w, b, A.linear_spec(rows), Cp, None)
                scale, shift = A.bn_eval_scale_shift(bn)
                if Cp != Cout:
                    scale, shift = F.pad(scale, (0, Cp - Cout)), F.pad(shift, (0, Cp - Cout))
                x = A._BNEvalActFn.apply(x, scale.contiguous(), shift.contiguous(), None, 1)
        else:
            x = torch.relu(A.conv(x, w, b, A.linear_spec(rows), Cp, None))
        return _PfnPoolFn.apply(x, M, P, self.last_vfe)


class PillarVFE(VFETemplate):
    def __init__(self, model_cfg, num_point_features, voxel_size, point_cloud_range, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.use_norm = self.model_cfg.USE_NORM
        self.with_distance = self.model_cfg.WITH_DISTANCE
        self.use_absolute_xyz = self.model_cfg.USE_ABSLOTE_XYZ
        self.num_raw_features = num_point_features
        num_point_features += 6 if self.use_absolute_xyz else 3
        if self.with_distance:
            num_point_features += 1
        self.num_filters = self.model_cfg.NUM_FILTERS
        assert len(self.num_filters) > 0
        num_filters = [num_point_features] + list(self.num_filters)
        self.pfn_layers = nn.ModuleList([PFNLayer(num_filters[i], num_filters[i + 1], self.use_norm, last_layer=(i >= len(num_filters) - 2))
                                         for i in range(len(num_filters) - 1)])
        self.voxel_x, self.voxel_y, self.voxel_z = voxel_size[0], voxel_size[1], voxel_size[2]
        self.x_offset = self.voxel_x / 2 + point_cloud_range[0]
        self.y_offset = self.voxel_y / 2 + point_cloud_range[1]
        self.z_offset = self.voxel_z / 2 + point_cloud_range[2]

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    def forward(self, batch_dict, **kwargs):
        voxels = batch_dict['voxels'].float().contiguous()
        num = batch_dict['voxel_num_points'].int().contiguous()
        coords = batch_dict['voxel_coords'].int().contiguous()
        geom = (self.voxel_x, self.voxel_y, self.voxel_z, self.x_offset, self.y_offset, self.z_offset)
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        synced = self.use_norm and self.training and A.sync_group(self.pfn_layers[0].norm) is not None
        if len(self.pfn_layers) == 1 and self.use_norm and not needs_grad and not synced and self.num_filters[0] <= 64:
            batch_dict['pillar_features'] = self._forward_fused(voxels, num, coords, geom)
            return batch_dict
        M, P = voxels.shape[0], voxels.shape[1]
        Cin = self.pfn_layers[0].linear.in_features
        x = K.pillar_decorate(voxels, num, coords, Cin, self.use_absolute_xyz, self.with_distance, geom, _pad32(Cin))
        cols = None
        for pfn in self.pfn_layers:
            x = pfn.forward_rows(x, M, P, cols)
            C, Cp = pfn.out_channels, _pad32(pfn.out_channels)
            # the next layer reads [x | max]: logical column c of either half sits at c / Cp + c of the padded concatenation
            cols = None if C == Cp else torch.cat([torch.arange(C, device=x.device), Cp + torch.arange(C, device=x.device)])
        C = self.pfn_layers[-1].out_channels
        batch_dict['pillar_features'] = x if x.shape[1] == C else x[:, :C]        # (M, Cout); the reference's .squeeze() of (M, 1, Cout)
        return batch_dict

    def _forward_fused(self, voxels, num, coords, geom):
        pfn = self.pfn_layers[0]
        w = pfn.linear.weight.detach().contiguous()
        bn = pfn.norm
        with torch.no_grad():
            if bn.training:
                M, P = voxels.shape[0], voxels.shape[1]
                stats = K.pillar_vfe_stats(voxels, num, coords, w, self.use_absolute_xyz, self.with_distance, geom)
                n = float(M * P)
                mean = stats[:w.shape[0]].double() / n
                var = (stats[w.shape[0]:].double() / n - mean * mean).clamp_min(0)
                rstd = torch.rsqrt(var + bn.eps)
                scale = (bn.weight.double() * rstd).float()
                shift = (bn.bias.double() - mean * bn.weight.double() * rstd).float()
                bn.running_mean.mul_(1 - bn.momentum).add_(bn.momentum * mean.float())
                bn.running_var.mul_(1 - bn.momentum).add_(bn.momentum * (var * n / max(n - 1, 1)).float())
                bn.num_batches_tracked += 1
            else:
                rstd = torch.rsqrt(bn.running_var + bn.eps)
                scale = (bn.weight * rstd).contiguous()
                shift = (bn.bias - bn.running_mean * scale).contiguous()
            return K.pillar_vfe_max(voxels, num, coords, w, self.use_absolute_xyz, self.with_distance, geom, scale.contiguous(), shift.contiguous())
