"""Radar_Distill: CMA densifier + student DenseEnc + AFD / PFD distillation losses.

Follows the reference's pcdet/models/backbones_2d/radar_distill_final.py:29-217 (module tree, batch_dict keys, loss
weights, tb_dict keys).  AFD (low_loss) of both radar maps is ONE fused HIP pass over the three BEV maps, PFD (high_loss)
one pass over the four maps (distill.hip) instead of ~45 elementwise kernels; tb_dict values stay device tensors
(the reference's 7 `.item()` syncs are deferred to whoever logs them).
"""
import torch
import torch.nn as nn

from radardistill_amd import autograd as A
from radardistill_amd import dense as D
from radardistill_amd import kernels as K
from ...ops.basicblock.Basicblock_convn import ConvNeXtBlock
from .base_bev_backbone import BaseBEVBackboneV2


def clip_sigmoid(x, eps=1e-4):
    return torch.clamp(x.sigmoid(), min=eps, max=1 - eps)


class _AFDFn(torch.autograd.Function):
    """out[4] = (feature_a, mask_a, feature_b, mask_b) for radar maps a, b against one lidar map."""

    @staticmethod
    def forward(ctx, lidar_rows, ra_rows, rb_rows, batch):
        out, coef, rowinfo = K.afd_fwd(lidar_rows, ra_rows, rb_rows, batch)
        ctx.save_for_backward(lidar_rows, ra_rows, rb_rows, coef, rowinfo)
        return out

    @staticmethod
    def backward(ctx, g):
        lidar_rows, ra, rb, coef, rowinfo = ctx.saved_tensors
        ga, gb = K.afd_bwd(lidar_rows, ra, rb, rowinfo, coef, g.contiguous())
        return None, ga, gb, None


class _PFDFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, r1, l1, r2, l2, gt_hm, hm_logits):
        out, cls, counts = K.pfd_fwd(r1, l1, r2, l2, gt_hm, hm_logits)
        ctx.save_for_backward(r1, l1, r2, l2, cls, counts)
        return out

    @staticmethod
    def backward(ctx, g):
        r1, l1, r2, l2, cls, counts = ctx.saved_tensors
        g1, g2 = K.pfd_bwd(r1, l1, r2, l2, cls, counts, g.contiguous())
        return g1, None, g2, None, None, None


def _rows(x):
    return A.nchw_to_rows(x)[0]


class Radar_Distill(BaseBEVBackboneV2):
    def __init__(self, model_cfg, **kwargs):
        super().__init__(model_cfg, **kwargs)
        self.model_cfg = model_cfg
        for i in (1, 2, 3):
            setattr(self, f"encoder_{i}", nn.Sequential(ConvNeXtBlock(dim=256, downsample=True), ConvNeXtBlock(dim=256, downsample=False)))
            setattr(self, f"decoder_{i}", nn.Sequential(nn.ConvTranspose2d(256, 256, 4, 2, 1), nn.BatchNorm2d(256), nn.GELU()))
            setattr(self, f"agg_{i}", nn.Sequential(nn.Conv2d(512, 256, 1, 1, 0), nn.BatchNorm2d(256), nn.GELU()))
        # keep the reference's registration order encoder_1, decoder_1, agg_1, encoder_2, ... (state_dict order only)
        self.voxel_size = self.model_cfg.VOXEL_SIZE
        self.point_cloud_range = self.model_cfg.POINT_CLOUD_RANGE

    # ---- losses
    def low_loss_pair(self, lidar_bev, radar_a, radar_b):
        """AFD for two radar maps at once -> (feature_a, mask_a, feature_b, mask_b) tensor[4]."""
        B = radar_a.shape[0]
        return _AFDFn.apply(_rows(lidar_bev).detach(), _rows(radar_a), _rows(radar_b), B)

    def low_loss(self, lidar_bev, radar_bev):
        out = self.low_loss_pair(lidar_bev, radar_bev, radar_bev.detach())
        return out[0], out[1]

    def high_loss(self, radar_bev, radar_bev2, lidar_bev, lidar_bev2, heatmaps, radar_preds):
        gt = torch.cat(heatmaps, dim=1)
        pr = torch.cat([p['hm'] for p in radar_preds], dim=1).detach()
        n_hm = gt.shape[1]
        gt_rows = gt.permute(0, 2, 3, 1).reshape(-1, n_hm).contiguous()
        pr_rows = pr.permute(0, 2, 3, 1).reshape(-1, n_hm).contiguous()
        return _PFDFn.apply(_rows(radar_bev), _rows(lidar_bev).detach(), _rows(radar_bev2), _rows(lidar_bev2).detach(), gt_rows, pr_rows)[0]

    def get_loss(self, batch_dict):
        low_lidar_bev = batch_dict['multi_scale_2d_features']['x_conv4']
        low_radar_bev = batch_dict['radar_multi_scale_2d_features']['radar_spatial_features_8x_2']
        low_radar_de_8x = batch_dict['radar_multi_scale_2d_features']['radar_spatial_features_8x_1']
        afd = self.low_loss_pair(low_lidar_bev, low_radar_bev, low_radar_de_8x)
        feature_loss, mask_loss, de_8x_feature_loss, de_8x_mask_loss = afd[0], afd[1], afd[2], afd[3]
        high = self.high_loss(batch_dict['radar_spatial_features_2d'], batch_dict['radar_spatial_features_2d_8x'],
                              batch_dict['spatial_features_2d'], batch_dict['spatial_features_2d_8x'],
                              batch_dict['target_dicts']['heatmaps'], batch_dict['radar_pred_dicts'])
        high_distill_loss = high * 25
        low_distill_loss = (0.5 * (feature_loss + de_8x_feature_loss) + 0.5 * (mask_loss + de_8x_mask_loss)) * 5
        distill_loss = low_distill_loss + high_distill_loss
        tb_dict = {
            'low_feature_loss': low_distill_loss.detach(), 'high_distill_loss': high_distill_loss.detach(),
            'distll_loss': distill_loss.detach(), 'low_distill_de_8x_loss': de_8x_feature_loss.detach(),
            'low_distill_loss': feature_loss.detach(), 'mask_loss': mask_loss.detach(), 'mask_de_8x_loss': de_8x_mask_loss.detach(),
        }
        return distill_loss, tb_dict

    # ---- forward
    def _dec(self, seq, x):
        return D.conv_bn_act(x, seq[0], seq[1], None, act=2)

    def forward(self, data_dict):
        ms = data_dict['radar_multi_scale_2d_features']
        spatial_features = ms['x_conv4']
        en_16x = self.encoder_1(spatial_features)
        de_8x = self._dec(self.agg_1, A.cat_channels(self._dec(self.decoder_1, en_16x), spatial_features))
        en_32x = self.encoder_2(en_16x)
        de_16x = self._dec(self.agg_2, A.cat_channels(self._dec(self.decoder_2, en_32x), self.encoder_3(de_8x)))
        x_conv4 = self._dec(self.agg_3, A.cat_channels(self._dec(self.decoder_3, de_16x), de_8x))
        ms['radar_spatial_features_8x_2'] = x_conv4
        ms['radar_spatial_features_8x_1'] = de_8x
        up, feat = self.dense_enc(x_conv4, ms['x_conv5'])
        data_dict['radar_spatial_features_2d_8x'] = up
        data_dict['radar_spatial_features_2d'] = feat
        return data_dict
