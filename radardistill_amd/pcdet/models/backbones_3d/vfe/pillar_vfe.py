"""PillarVFE over the padded-voxel input format (pcdet/models/backbones_3d/vfe/pillar_vfe.py:8-123) on the HIP kernels of
hardvox.hip.  Module tree and state_dict names are the reference's (`pfn_layers.0.linear.weight`, `pfn_layers.0.norm.*`).
Inference / no-grad forward only: the RadarDistill training configs use the dynamic VFE; one PFN layer."""
import torch
import torch.nn as nn

from radardistill_amd import kernels as K
from .vfe_template import VFETemplate


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
        self.part = 50000


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
        if len(self.pfn_layers) != 1 or not self.use_norm:
            raise NotImplementedError("PillarVFE on MI355X: one PFN layer with BatchNorm (NUM_FILTERS: [64], USE_NORM: True)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("PillarVFE backward is not built: the RadarDistill training path uses the dynamic VFE "
                                      "(run under torch.no_grad() or freeze the module)")
        voxels = batch_dict['voxels'].float().contiguous()
        num = batch_dict['voxel_num_points'].int().contiguous()
        coords = batch_dict['voxel_coords'].int().contiguous()
        pfn = self.pfn_layers[0]
        w = pfn.linear.weight.detach().contiguous()
        geom = (self.voxel_x, self.voxel_y, self.voxel_z, self.x_offset, self.y_offset, self.z_offset)
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
            feats = K.pillar_vfe_max(voxels, num, coords, w, self.use_absolute_xyz, self.with_distance, geom, scale.contiguous(), shift.contiguous())
        batch_dict['pillar_features'] = feats        # (M, Cout); the reference's .squeeze() of (M, 1, Cout)
        return batch_dict
