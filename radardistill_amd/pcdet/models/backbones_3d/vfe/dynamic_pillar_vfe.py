"""Dynamic pillar VFE on the MI355X kernels.

Same class names, constructor arguments, batch_dict keys and state_dict names as the reference's
pcdet/models/backbones_3d/vfe/dynamic_pillar_vfe.py:146-313 (DynamicPillarVFESimple2D,
Radar_DynamicPillarVFESimple2D) and :14-46 (PFNLayerV2), but the ~25 ATen kernels + torch.unique sort +
torch_scatter atomics of the reference become: voxelise into a rank grid (no sort), per-pillar mean, and ONE fused
Linear -> BatchNorm -> ReLU -> per-pillar max kernel (two passes in training, for the batch statistics).
"""
import os

import torch
import torch.nn as nn

from radardistill_amd import autograd as A
from radardistill_amd import kernels as K
from radardistill_amd import sparse as SP
from .vfe_template import VFETemplate


class PFNLayerV2(nn.Module):
    """Parameter container with the reference's names (linear.weight, norm.*); the arithmetic runs in vfe.hip."""

    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        if not (use_norm and last_layer and out_channels == 32):
            raise NotImplementedError("the RadarDistill configs use one PFN layer: Linear(no bias)+BN+ReLU -> 32")
        self.last_vfe, self.use_norm = last_layer, use_norm
        self.linear = nn.Linear(in_channels, out_channels, bias=False)
        self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)
        self.relu = nn.ReLU()


class _PillarVFEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, gamma, beta, points, point_row, coords, acc, geom, scale, shift, mean, rstd, n_pillars, n_valid, order=None,
                offsets=None, sync=None):
        need_grad = mean is not None
        if order is not None:          # segmented path: the per-pillar sums come out of the same pass
            out, argmax, acc = K.vfe_seg_max(points, order, offsets, coords, weight.detach().contiguous(), geom, scale, shift, n_pillars, need_grad)
        else:
            out, argmax = K.vfe_linear_bn_relu_max(points, point_row, coords, acc, weight.detach(), geom, scale, shift, n_pillars, need_grad)
        if need_grad:
            ctx.save_for_backward(weight, gamma, beta, points, point_row, coords, acc, geom, mean, rstd, argmax)
            ctx.n_valid = n_valid
            ctx.sync = sync
        return out

    @staticmethod
    def backward(ctx, grad_out):
        weight, gamma, beta, points, point_row, coords, acc, geom, mean, rstd, argmax = ctx.saved_tensors
        gw, gg, gb = K.vfe_backward(points, point_row, coords, acc, weight.detach(), geom, mean, rstd, gamma.detach(), beta.detach(),
                                    grad_out.contiguous(), argmax, ctx.n_valid, sync=ctx.sync)
        return (gw, gg, gb) + (None,) * 14


class DynamicPillarVFESimple2D(VFETemplate):
    POINTS_KEY = "points"
    OUT_PREFIX = ""

    def __init__(self, model_cfg, num_point_features, voxel_size, grid_size, point_cloud_range, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.use_norm = self.model_cfg.USE_NORM
        self.with_distance = self.model_cfg.WITH_DISTANCE
        self.use_absolute_xyz = self.model_cfg.USE_ABSLOTE_XYZ
        self.use_cluster_xyz = self.model_cfg.get('USE_CLUSTER_XYZ', True)
        self.use_relative_xyz = self.model_cfg.get('USE_RELATIVE_XYZ', True)
        if self.with_distance or not (self.use_absolute_xyz and self.use_cluster_xyz and self.use_relative_xyz):
            raise NotImplementedError("HIP VFE implements the distill config: absolute+cluster+relative xyz, no distance")
        self.raw_point_features = int(num_point_features)
        num_point_features = num_point_features + 9
        self.num_point_features = num_point_features
        self.num_filters = list(self.model_cfg.NUM_FILTERS)
        if len(self.num_filters) != 1:
            raise NotImplementedError("single PFN layer only (NUM_FILTERS: [32])")
        self.pfn_layers = nn.ModuleList([PFNLayerV2(num_point_features, self.num_filters[0], self.use_norm, last_layer=True)])
        self.voxel_x, self.voxel_y, self.voxel_z = [float(v) for v in voxel_size]
        pcr = [float(v) for v in point_cloud_range]
        self.x_offset = self.voxel_x / 2 + pcr[0]
        self.y_offset = self.voxel_y / 2 + pcr[1]
        self.z_offset = self.voxel_z / 2 + pcr[2]
        self.grid_x, self.grid_y = int(grid_size[0]), int(grid_size[1])
        self.pc_range = pcr
        # [vx, vy, vz, x_off, y_off, z_off, x0, y0, z0] in fp32, as the reference's python-float * fp32-tensor arithmetic sees them
        self.register_buffer("_geom", torch.tensor([self.voxel_x, self.voxel_y, self.voxel_z, self.x_offset, self.y_offset,
                                                    self.z_offset, pcr[0], pcr[1], pcr[2]], dtype=torch.float32), persistent=False)
        if model_cfg.get("DOUBLE_FLIP", False):
            raise NotImplementedError("DOUBLE_FLIP test-time augmentation is outside the training hot path")

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    def _points(self, batch_dict):
        points = batch_dict[self.POINTS_KEY]
        if points.dtype != torch.float32 or not points.is_contiguous():
            points = points.float().contiguous()
        if points.shape[1] != 1 + self.raw_point_features:
            raise RuntimeError(f"{self.POINTS_KEY}: expected {1 + self.raw_point_features} columns, got {points.shape[1]}")
        return points

    def geometry_begin(self, batch_dict):
        """Index half of the VFE, part 1: points -> rank grid + point->pillar rows.  Returns the launch state and the two device
        scalars (pillar count, in-range point count) the host needs; detectors/pillarnet.py reads them for all branches at once."""
        points = self._points(batch_dict)
        B = int(batch_dict['batch_size'])
        rg, point_row = K.voxelize(points, B, self.grid_x, self.grid_y, self.pc_range[0], self.pc_range[1], self.voxel_x, self.voxel_y)
        cnt = K.rankgrid_count_tensor(rg, B * self.grid_x * self.grid_y)
        return (points, rg, point_row, B), [cnt.long(), (point_row >= 0).sum()]

    def geometry_finish(self, batch_dict, state, P, n_valid):
        """Part 2: pillar coordinates, registered with their rank grid (and a `_Level` the sparse backbone will adopt)."""
        points, rg, point_row, B = state
        coords = K.rankgrid_coords(rg, B, self.grid_y, self.grid_x, True, P)     # (b, y, x), rows in (b, cx, cy) key order
        level = SP._Level(coords, rg, True, B, self.grid_y, self.grid_x)
        SP.register_rankgrid(coords, rg, True, level)
        batch_dict[self.OUT_PREFIX + '_vfe_geometry'] = (points, rg, point_row, coords, P, n_valid)
        return level

    def prelude_begin(self, batch_dict, n_down, scalars):
        """geometry_begin + the rank grids of the `n_down` encoder levels below the pillar grid in ONE library call
        (rd_geometry_begin); `scalars` (int32 device, 2 + n_down) receives the sizes the host reads."""
        points = self._points(batch_dict)
        B = int(batch_dict['batch_size'])
        rgs, point_row, dims = K.geometry_begin(points, B, self.grid_x, self.grid_y, self.pc_range[0], self.pc_range[1], self.voxel_x,
                                                self.voxel_y, n_down, scalars)
        return points, rgs, point_row, dims, B

    def prelude_finish(self, batch_dict, state, vals):
        """vals = [pillars, in-range points, rows of each level below]: coordinates + all neighbour tables of the branch in ONE
        library call (rd_geometry_finish); returns the pillar `_Level` with its pyramid attached."""
        points, rgs, point_row, dims, B = state
        P, n_valid = int(vals[0]), int(vals[1])
        coords, subm, down, up = K.geometry_finish(rgs, B, self.grid_x, self.grid_y, [P, *vals[2:]])
        level = SP.pyramid_from_tables(rgs, dims, B, coords, subm, down, up)
        SP.register_rankgrid(coords[0], rgs[0], True, level)
        batch_dict[self.OUT_PREFIX + '_vfe_geometry'] = (points, rgs[0], point_row, coords[0], P, n_valid)
        return level

    def forward(self, batch_dict, **kwargs):
        geo = batch_dict.pop(self.OUT_PREFIX + '_vfe_geometry', None)
        if geo is None:
            state, scalars = self.geometry_begin(batch_dict)
            P, n_valid = [int(v) for v in torch.stack(scalars).tolist()]          # one device->host sync
            self.geometry_finish(batch_dict, state, P, n_valid)
            geo = batch_dict.pop(self.OUT_PREFIX + '_vfe_geometry')
        points, rg, point_row, coords, P, n_valid = geo
        B = int(batch_dict['batch_size'])
        g = self._geom
        pfn = self.pfn_layers[0]
        w, bn = pfn.linear.weight, pfn.norm
        # default: points grouped by pillar, one wavefront per pillar, shuffle reductions, no float atomics (vfe_seg.hip);
        # RD_VFE_SEG=0: the first version (per-point lanes + 64-bit atomicMax into a packed buffer)
        if P == 0 and bn.training and A.sync_group(bn) is not None:
            raise RuntimeError("SyncBatchNorm VFE: this rank's batch has no in-range point; every rank must contribute to the statistics")
        seg = P > 0 and os.environ.get("RD_VFE_SEG", "1") != "0"
        if seg:
            return self._forward_segmented(batch_dict, points, point_row, coords, P, n_valid, g, w, bn)
        acc = K.vfe_pillar_mean(points, point_row, P)
        if P == 0:
            feats = points.new_zeros((0, 32))
        elif bn.training:
            stats = K.vfe_linear_stats(points, point_row, coords, acc, w.detach().contiguous(), g)
            sync = _sync_stats(bn, stats)
            if n_valid <= 1 and sync is None:
                raise ValueError("Expected more than 1 value per channel when training")
            A._BN_TOUCHED.append(bn)
            mean, rstd, scale, shift = K.bn_finalize(stats, n_valid, 32, bn.weight.detach(), bn.bias.detach(), float(bn.eps),
                                                     float(bn.momentum), bn.running_mean, bn.running_var, sync=sync is not None)
            if torch.is_grad_enabled() and w.requires_grad:
                feats = _PillarVFEFn.apply(w, bn.weight, bn.bias, points, point_row, coords, acc, g, scale, shift, mean, rstd, P, n_valid,
                                           None, None, sync)
            else:
                feats, _ = K.vfe_linear_bn_relu_max(points, point_row, coords, acc, w.detach().contiguous(), g, scale, shift, P, False)
        else:
            rstd = torch.rsqrt(bn.running_var + bn.eps)
            scale = (bn.weight * rstd).detach().contiguous()
            shift = (bn.bias - bn.running_mean * bn.weight * rstd).detach().contiguous()
            feats, _ = K.vfe_linear_bn_relu_max(points, point_row, coords, acc, w.detach().contiguous(), g, scale, shift, P, False)
        batch_dict[self.OUT_PREFIX + 'pillar_features'] = feats
        batch_dict[self.OUT_PREFIX + 'pillar_coords'] = coords
        return batch_dict


def _sync_stats(bn, stats):
    """SyncBatchNorm (--sync_bn): the 65-value statistics buffer (sums, sums of squares, valid-point count) is summed over the
    process group in place -> (all-reduce callable, device count) for the backward, or None when this BatchNorm is local."""
    group = A.sync_group(bn)
    if group is None:
        return None
    reduce = A._group_sum(group)
    reduce(stats)
    return reduce, stats[64:]


def _segmented(self, batch_dict, points, point_row, coords, P, n_valid, g, w, bn):
    offsets, order = K.vfe_group(point_row, P)
    wd = w.detach().contiguous()
    if bn.training:
        stats = K.vfe_seg_stats(points, order, offsets, coords, wd, g, P)
        sync = _sync_stats(bn, stats)
        if n_valid <= 1 and sync is None:
            raise ValueError("Expected more than 1 value per channel when training")
        A._BN_TOUCHED.append(bn)
        mean, rstd, scale, shift = K.bn_finalize(stats, n_valid, 32, bn.weight.detach(), bn.bias.detach(), float(bn.eps),
                                                 float(bn.momentum), bn.running_mean, bn.running_var, sync=sync is not None)
        if torch.is_grad_enabled() and w.requires_grad:
            feats = _PillarVFEFn.apply(w, bn.weight, bn.bias, points, point_row, coords, None, g, scale, shift, mean, rstd, P, n_valid, order, offsets,
                                       sync)
        else:
            feats, _, _ = K.vfe_seg_max(points, order, offsets, coords, wd, g, scale, shift, P, False)
    else:
        rstd = torch.rsqrt(bn.running_var + bn.eps)
        scale = (bn.weight * rstd).detach().contiguous()
        shift = (bn.bias - bn.running_mean * bn.weight * rstd).detach().contiguous()
        feats, _, _ = K.vfe_seg_max(points, order, offsets, coords, wd, g, scale, shift, P, False)
    batch_dict[self.OUT_PREFIX + 'pillar_features'] = feats
    batch_dict[self.OUT_PREFIX + 'pillar_coords'] = coords
    return batch_dict


DynamicPillarVFESimple2D._forward_segmented = _segmented


class Radar_DynamicPillarVFESimple2D(DynamicPillarVFESimple2D):
    """Reads `radar_points`, writes `radar_pillar_features` / `radar_pillar_coords` (dynamic_pillar_vfe.py:255-313)."""
    POINTS_KEY = "radar_points"
    OUT_PREFIX = "radar_"


class Radar_DynamicPillarVFESimple2D_Test(DynamicPillarVFESimple2D):
    """The eval-graph radar VFE of radar_distill_val.yaml:67 (dynamic_pillar_vfe.py:315-375): the radar-only test dataset hands its
    sweep over as `points`, the outputs go to the student's `radar_pillar_*` keys; state_dict names are the base class's."""
    POINTS_KEY = "points"
    OUT_PREFIX = "radar_"
