"""PointPillarScatter (pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py:5-37): pillar rows -> dense BEV map, one
scatter launch for the whole batch (the reference loops over samples with boolean masks and a host read of the batch size);
the map is channels-last memory behind a logical (B, C, ny, nx) tensor."""
import torch
import torch.nn as nn

from radardistill_amd import autograd as A


class PointPillarScatter(nn.Module):
    def __init__(self, model_cfg, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = self.model_cfg.NUM_BEV_FEATURES
        self.nx, self.ny, self.nz = [int(v) for v in grid_size]
        assert self.nz == 1

    def forward(self, batch_dict, **kwargs):
        feats, coords = batch_dict['pillar_features'], batch_dict['voxel_coords']
        batch_size = int(batch_dict['batch_size']) if 'batch_size' in batch_dict else int(coords[:, 0].max().item()) + 1
        byx = coords[:, [0, 2, 3]].int().contiguous()               # (b, z, y, x) -> (b, y, x); nz == 1
        rows = A.rows_to_dense(feats.contiguous(), byx, batch_size, self.ny, self.nx)
        batch_dict['spatial_features'] = A.rows_to_nchw(rows, batch_size, self.ny, self.nx)
        return batch_dict
