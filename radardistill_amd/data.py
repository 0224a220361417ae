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
        """Geometry from a loaded reference yaml.  VOXEL_SIZE lives in the DATA_PROCESSOR list (the last entry that carries one
        wins, as each processor overwrites `self.voxel_size`: data_processor.py:116-124,142-146,263-268;
        radar_distill_train.yaml:62-63); feature counts are the lengths of the POINT_FEATURE_ENCODING lists
        (point_feature_encoder.py:86-90)."""
        d = cfg.DATA_CONFIG
        voxel = d.get('VOXEL_SIZE', None)
        for proc in d.get('DATA_PROCESSOR', None) or []:
            if proc.get('VOXEL_SIZE', None) is not None:
                voxel = proc['VOXEL_SIZE']
        if voxel is None:
            raise KeyError("no VOXEL_SIZE in DATA_CONFIG.DATA_PROCESSOR (transform_points_to_voxels[_placeholder]) nor DATA_CONFIG")
        enc = d.get('POINT_FEATURE_ENCODING', None) or {}
        n_pts = len(enc['used_feature_list']) if 'used_feature_list' in enc else d.get('NUM_POINT_FEATURES', 5)
        n_rad = len(enc['radar_used_feature_list']) if 'radar_used_feature_list' in enc else d.get('RADAR_NUM_POINT_FEATURES', 6)
        return cls(cfg.CLASS_NAMES, d.POINT_CLOUD_RANGE, list(voxel), n_pts, n_rad)


def collate_batch(batch_list, _unused=False):
    """Sample dicts -> batch dict, as NuScenesDataset_Distill.collate_batch (pcdet/datasets/dataset_distill.py:220-325) for the keys
    of the distillation path: point / voxel-coordinate arrays get the sample index prepended as column 0 and are concatenated,
    padded-voxel payloads are concatenated, `gt_boxes` is zero-padded to the longest sample, everything else is stacked.
    Lists of arrays per sample (the DOUBLE_FLIP test-time augmentation) multiply `batch_size` by their length."""
    from collections import defaultdict
    data = defaultdict(list)
    for sample in batch_list:
        for k, v in sample.items():
            data[k].append(v)
    batch_size, ratio = len(batch_list), 1
    ret = {}
    for key, val in data.items():
        if key in ('voxels', 'voxel_num_points', 'radar_voxels', 'radar_voxel_num_points'):
            if isinstance(val[0], list):
                ratio = len(val[0])
                val = [i for item in val for i in item]
            ret[key] = np.concatenate(val, axis=0)
        elif key in ('points', 'voxel_coords', 'radar_points', 'radar_voxel_coords'):
            if isinstance(val[0], list):
                val = [i for item in val for i in item]
            ret[key] = np.concatenate([np.pad(c, ((0, 0), (1, 0)), mode='constant', constant_values=i) for i, c in enumerate(val)], axis=0)
        elif key == 'gt_boxes':
            max_gt = max(len(x) for x in val)
            out = np.zeros((batch_size, max_gt, val[0].shape[-1]), dtype=np.float32)
            for k in range(batch_size):
                out[k, :len(val[k]), :] = val[k]
            ret[key] = out
        elif key == 'calib':
            ret[key] = val
        else:
            ret[key] = np.stack(val, axis=0)
    ret['batch_size'] = batch_size * ratio
    return ret


class SyntheticSweeps(SyntheticDistillDataset):
    """Map-style dataset of synthetic nuScenes-shaped sweeps (SURVEY 8(d) distributions) yielding the reference's per-sample dicts
    (`points` (n,5), `radar_points` (m,6), `gt_boxes` (k,10), `frame_id`); use with torch DataLoader(collate_fn=collate_batch)."""

    def __init__(self, length, grid=512, n_lidar=35000, n_radar=2000, n_boxes=30, seed=0, **kw):
        from .synthetic import bench_geometry
        pc_range, voxel, _ = bench_geometry(grid)
        super().__init__(kw.pop('class_names', ['car', 'truck', 'construction_vehicle', 'bus', 'trailer', 'barrier', 'motorcycle', 'bicycle',
                                                'pedestrian', 'traffic_cone']), pc_range, voxel, **kw)
        self.length, self.grid, self.n_lidar, self.n_radar, self.n_boxes, self.seed = length, grid, n_lidar, n_radar, n_boxes, seed

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        from .synthetic import make_batch
        b = make_batch(batch_size=1, n_lidar=self.n_lidar, n_radar=self.n_radar, n_boxes=self.n_boxes, grid=self.grid, seed=self.seed + index)
        return {'points': b['points'][:, 1:], 'radar_points': b['radar_points'][:, 1:], 'gt_boxes': b['gt_boxes'][0], 'frame_id': np.int64(index)}
