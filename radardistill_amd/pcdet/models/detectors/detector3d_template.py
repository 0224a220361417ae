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

    # ---- checkpoints (detector3d_template.py:411-496)
    @staticmethod
    def generate_recall_record(box_preds, recall_dict, batch_index, data_dict=None, thresh_list=None):
        """detector3d_template.py:367-409: how many ground-truth boxes have a prediction above each 3-D IoU threshold (the pairwise
        IoU runs on the HIP overlap kernel; one host read per threshold, as in the reference)."""
        from ...ops.iou3d_nms import iou3d_nms_utils
        if 'gt_boxes' not in data_dict:
            return recall_dict
        rois = data_dict['rois'][batch_index] if 'rois' in data_dict else None
        gt_boxes = data_dict['gt_boxes'][batch_index]
        if recall_dict.__len__() == 0:
            recall_dict = {'gt': 0}
            for cur_thresh in thresh_list:
                recall_dict['roi_%s' % (str(cur_thresh))] = 0
                recall_dict['rcnn_%s' % (str(cur_thresh))] = 0
        cur_gt = gt_boxes
        nonzero = (cur_gt.sum(dim=1) != 0).nonzero()
        k = int(nonzero.max().item()) if nonzero.numel() else -1          # trailing all-zero rows are padding
        cur_gt = cur_gt[:k + 1]
        if cur_gt.shape[0] > 0:
            if box_preds.shape[0] > 0:
                iou3d_rcnn = iou3d_nms_utils.boxes_iou3d_gpu(box_preds[:, 0:7], cur_gt[:, 0:7])
            else:
                iou3d_rcnn = torch.zeros((0, cur_gt.shape[0]))
            if rois is not None:
                iou3d_roi = iou3d_nms_utils.boxes_iou3d_gpu(rois[:, 0:7], cur_gt[:, 0:7])
            for cur_thresh in thresh_list:
                if iou3d_rcnn.shape[0] > 0:
                    recall_dict['rcnn_%s' % str(cur_thresh)] += (iou3d_rcnn.max(dim=0)[0] > cur_thresh).sum().item()
                if rois is not None:
                    recall_dict['roi_%s' % str(cur_thresh)] += (iou3d_roi.max(dim=0)[0] > cur_thresh).sum().item()
            recall_dict['gt'] += cur_gt.shape[0]
        return recall_dict

    def _load_state_dict(self, model_state_disk, *, strict=True):
        state_dict = self.state_dict()
        spconv_keys = find_all_spconv_keys(self)
        update_model_state = {}
        for key, val in model_state_disk.items():
            if key in spconv_keys and key in state_dict and state_dict[key].shape != val.shape:
                # spconv 1.x stored (k1, k2, c_in, c_out); 2.x / this build store (c_out, k1, k2, c_in)
                val_native = val.transpose(-1, -2)
                if val_native.shape == state_dict[key].shape:
                    val = val_native.contiguous()
                elif val.dim() == 4:
                    val_implicit = val.permute(3, 0, 1, 2)
                    if val_implicit.shape == state_dict[key].shape:
                        val = val_implicit.contiguous()
            if key in state_dict and state_dict[key].shape == val.shape:
                update_model_state[key] = val
        if strict:
            self.load_state_dict(update_model_state)
        else:
            state_dict.update(update_model_state)
            self.load_state_dict(state_dict)
        return state_dict, update_model_state

    def load_params_from_file(self, filename, logger, to_cpu=False, pre_trained_path=None):
        if not os.path.isfile(filename):
            raise FileNotFoundError
        logger.info('==> Loading parameters from checkpoint %s to %s' % (filename, 'CPU' if to_cpu else 'GPU'))
        loc_type = torch.device('cpu') if to_cpu else None
        checkpoint = torch.load(filename, map_location=loc_type, weights_only=True)
        model_state_disk = checkpoint['model_state']
        if pre_trained_path is not None:
            model_state_disk.update(torch.load(pre_trained_path, map_location=loc_type, weights_only=True)['model_state'])
        version = checkpoint.get("version", None)
        if version is not None:
            logger.info('==> Checkpoint trained from version: %s' % version)
        state_dict, update_model_state = self._load_state_dict(model_state_disk, strict=False)
        for key in state_dict:
            if key not in update_model_state:
                logger.info('Not updated weight %s: %s' % (key, str(state_dict[key].shape)))
        logger.info('==> Done (loaded %d/%d)' % (len(update_model_state), len(state_dict)))

    def load_params_with_optimizer(self, filename, to_cpu=False, optimizer=None, logger=None):
        if not os.path.isfile(filename):
            raise FileNotFoundError
        loc_type = torch.device('cpu') if to_cpu else None
        checkpoint = torch.load(filename, map_location=loc_type, weights_only=True)
        epoch = checkpoint.get('epoch', -1)
        it = checkpoint.get('it', 0.0)
        self._load_state_dict(checkpoint['model_state'], strict=True)
        if optimizer is not None:
            if checkpoint.get('optimizer_state', None) is not None:
                optimizer.load_state_dict(checkpoint['optimizer_state'])
            else:
                # the moments may sit next to the checkpoint as `<name>_optim.<ext>` (detector3d_template.py:484-490)
                assert filename[-4] == '.', filename
                optimizer_filename = '%s_optim.%s' % (filename[:-4], filename[-3:])
                if os.path.exists(optimizer_filename):
                    optimizer_ckpt = torch.load(optimizer_filename, map_location=loc_type, weights_only=True)
                    optimizer.load_state_dict(optimizer_ckpt['optimizer_state'])
        return it, epoch
