"""GPU voxel generator of the padded-voxel input format: the role of `VoxelGeneratorWrapper` / `transform_points_to_voxels`
(pcdet/datasets/processor/data_processor.py:16-61,142-229), moved from the CPU data workers onto the device and batched.

    gen = VoxelGenerator(vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels)
    voxels, coords, num_points = gen.generate(points)      # points (N, 1 + C) CUDA, batch id first, sorted by batch id
    batch_dict.update(voxels=voxels, voxel_coords=coords, voxel_num_points=num_points)   # coords (M, 4) = (b, z, y, x)
"""
import numpy as np

from . import kernels as K


class VoxelGenerator:
    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels):
        self.vsize = [float(v) for v in vsize_xyz]
        self.range = [float(v) for v in coors_range_xyz]
        self.num_point_features = int(num_point_features)
        self.max_points = int(max_num_points_per_voxel)
        self.max_voxels = int(max_num_voxels)
        grid = (np.array(self.range[3:6]) - np.array(self.range[0:3])) / np.array(self.vsize)
        self.grid_size = np.round(grid).astype(np.int64)          # data_processor.py:144-145

    def generate(self, points, batch_size=None):
        if points.shape[1] != 1 + self.num_point_features:
            raise RuntimeError(f"points have {points.shape[1] - 1} features, generator was built for {self.num_point_features}")
        if batch_size is None:
            batch_size = int(points[-1, 0].item()) + 1 if points.shape[0] else 1
        return K.voxelize_hard(points.float().contiguous(), int(batch_size), self.grid_size, self.range, self.vsize, self.max_points, self.max_voxels)
