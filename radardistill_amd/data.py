"""Dataset-side contract that `build_network` needs: class_names, point_feature_encoder.{num_point_features,
radar_num_point_features}, grid_size, point_cloud_range, voxel_size, depth_downsample_factor -- for synthetic sweeps.
The reference derives these in pcdet/datasets/dataset_distill.py + processor/data_processor.py:116-124
(transform_points_to_voxels_placeholder: grid_size = round((range_max - range_min) / voxel_size))."""
from types import SimpleNamespace

import numpy as np


class SyntheticDistillDataset:
    def __init__(self, class_names, point_cloud_range, voxel_size, num_point_features=5, radar_num_point_features=6):
        self.class_names = list(class_names)
        self.point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        self.voxel_size = list(voxel_size)
        grid = (self.point_cloud_range[3:6] - self.point_cloud_range[0:3]) / np.array(voxel_size)
        self.grid_size = np.round(grid).astype(np.int64)
        self.depth_downsample_factor = None
        self.point_feature_encoder = SimpleNamespace(num_point_features=num_point_features,
                                                     radar_num_point_features=radar_num_point_features)

    @classmethod
    def from_cfg(cls, cfg):
        d = cfg.DATA_CONFIG
        return cls(cfg.CLASS_NAMES, d.POINT_CLOUD_RANGE, d.VOXEL_SIZE, d.get('NUM_POINT_FEATURES', 5), d.get('RADAR_NUM_POINT_FEATURES', 6))
