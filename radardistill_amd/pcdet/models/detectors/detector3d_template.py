"""Detector3DTemplate: module registry walk, state-dict loading (spconv weight-layout fix-up), checkpoint I/O.
Public surface of pcdet/models/detectors/detector3d_template.py:14-496 restricted to the slots PillarNet uses."""
import os

import torch
import torch.nn as nn

from ...utils.spconv_utils import find_all_spconv_keys
from .. import backbones_2d, backbones_3d, dense_heads
from ..backbones_3d import vfe


class Detector3DTemplate(nn.Module):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.dataset = dataset
        self.class_names = dataset.class_names
        self.register_buffer('global_step', torch.LongTensor(1).zero_())
        self.module_topology = [
            'vfe', 'radar_vfe', 'backbone_3d', 'radar_backbone_3d', 'map_to_bev_module', 'radar_map_to_bev_module', 'pfe',
            'backbone_2d', 'radar_backbone_2d', 'dense_head', 'radar_dense_head', 'point_head', 'roi_head'
        ]

    @property
    def mode(self):
        return 'TRAIN' if self.training else 'TEST'

    def update_global_step(self):
        self.global_step += 1

    def build_networks(self):
        pfe = self.dataset.point_feature_encoder
        info = {
            'module_list': [],
            'num_rawpoint_features': pfe.num_point_features,
            'num_point_features': pfe.num_point_features,
            'grid_size': self.dataset.grid_size,
            'point_cloud_range': self.dataset.point_cloud_range,
            'voxel_size': self.dataset.voxel_size,
            'depth_downsample_factor': getattr(self.dataset, 'depth_downsample_factor', None),
        }
        if hasattr(pfe, 'radar_num_point_features'):
            info.update({'radar_num_rawpoint_features': pfe.radar_num_point_features,
                         'radar_num_point_features': pfe.radar_num_point_features})
        for module_name in self.module_topology:
            module, info = getattr(self, 'build_%s' % module_name)(model_info_dict=info)
            self.add_module(module_name, module)
        return info['module_list']

    # ---- slots used by PillarNet (same kwargs as the reference, detector3d_template.py:59-219)
    def _build_vfe(self, key, info, num_feat, out_key):
        cfg = self.model_cfg.get(key, None)
        if cfg is None:
            return None, info
        m = vfe.__all__[cfg.NAME](model_cfg=cfg, num_point_features=num_feat, point_cloud_range=info['point_cloud_range'],
                                  voxel_size=info['voxel_size'], grid_size=info['grid_size'],
                                  depth_downsample_factor=info['depth_downsample_factor'])
        info[out_key] = m.get_output_feature_dim()
        info['module_list'].append(m)
        return m, info

    def build_vfe(self, model_info_dict):
        return self._build_vfe('VFE', model_info_dict, model_info_dict['num_rawpoint_features'], 'num_point_features')

    def build_radar_vfe(self, model_info_dict):
        # the reference hard-codes 6 radar point features (detector3d_template.py:80)
        return self._build_vfe('RADAR_VFE', model_info_dict, 6, 'radar_num_point_features')

    def _build_b3d(self, key, info, in_key, ch_key):
        cfg = self.model_cfg.get(key, None)
        if cfg is None:
            return None, info
        m = backbones_3d.__all__[cfg.NAME](model_cfg=cfg, input_channels=info[in_key], grid_size=info['grid_size'],
                                           voxel_size=info['voxel_size'], point_cloud_range=info['point_cloud_range'])
        info['module_list'].append(m)
        info[in_key] = m.num_point_features
        info[ch_key] = getattr(m, 'backbone_channels', None)
        return m, info

    def build_backbone_3d(self, model_info_dict):
        return self._build_b3d('BACKBONE_3D', model_info_dict, 'num_point_features', 'backbone_channels')

    def build_radar_backbone_3d(self, model_info_dict):
        return self._build_b3d('RADAR_BACKBONE_3D', model_info_dict, 'radar_num_point_features', 'radar_backbone_channels')

    def _build_b2d(self, key, info, feat_key):
        cfg = self.model_cfg.get(key, None)
        if cfg is None:
            return None, info
        m = backbones_2d.__all__[cfg.NAME](model_cfg=cfg, input_channels=info.get(feat_key, None))
        info['module_list'].append(m)
        info[feat_key] = m.num_bev_features
        return m, info

    def build_backbone_2d(self, model_info_dict):
        return self._build_b2d('BACKBONE_2D', model_info_dict, 'num_bev_features')

    def build_radar_backbone_2d(self, model_info_dict):
        return self._build_b2d('RADAR_BACKBONE_2D', model_info_dict, 'radar_num_bev_features')

    def _build_head(self, key, info, feat_key):
        cfg = self.model_cfg.get(key, None)
        if cfg is None:
            return None, info
        m = dense_heads.__all__[cfg.NAME](
            model_cfg=cfg, input_channels=info[feat_key] if feat_key in info else cfg.INPUT_FEATURES,
            num_class=self.num_class if not cfg.CLASS_AGNOSTIC else 1, class_names=self.class_names,
            grid_size=info['grid_size'], point_cloud_range=info['point_cloud_range'],
            predict_boxes_when_training=self.model_cfg.get('ROI_HEAD', False), voxel_size=info.get('voxel_size', False))
        info['module_list'].append(m)
        return m, info

    def build_dense_head(self, model_info_dict):
        return self._build_head('DENSE_HEAD', model_info_dict, 'num_bev_features')

    def build_radar_dense_head(self, model_info_dict):
        return self._build_head('RADAR_DENSE_HEAD', model_info_dict, 'radar_num_bev_features')

    def _unsupported(self, key, info):
        if self.model_cfg.get(key, None) is not None:
            raise NotImplementedError(f"MODEL.{key} is outside the RadarDistill training hot path (SURVEY section 8)")
        return None, info

    def build_map_to_bev_module(self, model_info_dict):
        """detector3d_template.py:111-121: PointPillarScatter of the padded-voxel input format."""
        cfg = self.model_cfg.get('MAP_TO_BEV', None)
        if cfg is None:
            return None, model_info_dict
        from ..backbones_2d import map_to_bev
        m = map_to_bev.__all__[cfg.NAME](model_cfg=cfg, grid_size=model_info_dict['grid_size'])
        model_info_dict['module_list'].append(m)
        model_info_dict['num_bev_features'] = m.num_bev_features
        return m, model_info_dict

    def build_radar_map_to_bev_module(self, model_info_dict):
        return self._unsupported('RADAR_MAP_TO_BEV', model_info_dict)

    def build_pfe(self, model_info_dict):
        return self._unsupported('PFE', model_info_dict)

    def build_point_head(self, model_info_dict):
        return self._unsupported('POINT_HEAD', model_info_dict)

    def build_roi_head(self, model_info_dict):
        return self._unsupported('ROI_HEAD', model_info_dict)

    def forward(self, **kwargs):
        raise NotImplementedError

    # ---- recall records and checkpoints: the logic lives in radardistill_amd/checkpoint.py; these are the reference's entry points
    @staticmethod
    def generate_recall_record(box_preds, recall_dict, batch_index, data_dict=None, thresh_list=None):
        """detector3d_template.py:367-409 -- `recall_dict` counts, per 3-D IoU threshold, the ground-truth boxes that some prediction
        (`rcnn_<t>`) / some first-stage proposal (`roi_<t>`, only when the batch carries `rois`) overlaps by more than the threshold,
        and the ground-truth boxes seen (`gt`).  Pairwise IoU on the HIP overlap kernel."""
        from .... import checkpoint as CK
        from ...ops.iou3d_nms import iou3d_nms_utils
        if 'gt_boxes' not in data_dict:
            return recall_dict
        if not recall_dict:
            recall_dict = {'gt': 0}
            for t in thresh_list:
                recall_dict[f'roi_{t}'] = 0
                recall_dict[f'rcnn_{t}'] = 0
        gt = CK.strip_padding(data_dict['gt_boxes'][batch_index])
        if gt.shape[0] == 0:
            return recall_dict
        sources = {'rcnn': box_preds}
        if 'rois' in data_dict:
            sources['roi'] = data_dict['rois'][batch_index]
        for tag, boxes in sources.items():
            if boxes.shape[0] > 0:
                hits = CK.count_recalled(iou3d_nms_utils.boxes_iou3d_gpu(boxes[:, 0:7], gt[:, 0:7]), thresh_list)
                for t, n in zip(thresh_list, hits):
                    recall_dict[f'{tag}_{t}'] += n
        recall_dict['gt'] += gt.shape[0]
        return recall_dict

    def _load_state_dict(self, model_state_disk, *, strict=True):
        """-> (the model's state dict after loading, the entries taken from disk).  strict: exactly the fitted entries must make up the
        whole model (torch raises otherwise); non-strict: entries that do not fit keep the model's current values."""
        from .... import checkpoint as CK
        own, fitted = CK.fit_state_to_model(self, model_state_disk, find_all_spconv_keys(self))
        if strict:
            self.load_state_dict(fitted)
        else:
            own.update(fitted)
            self.load_state_dict(own)
        return own, fitted

    def load_params_from_file(self, filename, logger, to_cpu=False, pre_trained_path=None):
        from .... import checkpoint as CK
        checkpoint = CK.read_checkpoint(filename, to_cpu)
        logger.info('==> Loading parameters from checkpoint %s to %s' % (filename, 'CPU' if to_cpu else 'GPU'))
        disk = checkpoint['model_state']
        if pre_trained_path is not None:          # a second file overlays the first (detector3d_template.py:449-452)
            disk.update(CK.read_checkpoint(pre_trained_path, to_cpu)['model_state'])
        if checkpoint.get('version', None) is not None:
            logger.info('==> Checkpoint trained from version: %s' % checkpoint['version'])
        own, fitted = self._load_state_dict(disk, strict=False)
        for key in (k for k in own if k not in fitted):
            logger.info('Not updated weight %s: %s' % (key, str(own[key].shape)))
        logger.info('==> Done (loaded %d/%d)' % (len(fitted), len(own)))

    def load_params_with_optimizer(self, filename, to_cpu=False, optimizer=None, logger=None):
        """-> (it, epoch).  The optimizer state comes from the checkpoint or, failing that, from its `_optim` side file."""
        from .... import checkpoint as CK
        checkpoint = CK.read_checkpoint(filename, to_cpu)
        self._load_state_dict(checkpoint['model_state'], strict=True)
        if optimizer is not None:
            opt_state = checkpoint.get('optimizer_state', None)
            if opt_state is None and os.path.exists(CK.optimizer_side_file(filename)):
                opt_state = CK.read_checkpoint(CK.optimizer_side_file(filename), to_cpu)['optimizer_state']
            if opt_state is not None:
                optimizer.load_state_dict(opt_state)
        return checkpoint.get('it', 0.0), checkpoint.get('epoch', -1)
