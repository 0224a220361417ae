"""Student SparseEnc: same network, radar_* batch_dict keys
(pcdet/models/backbones_3d/spconv_backbone_2d_distillation.py:6-96)."""
from .spconv_backbone_2d import PillarRes18BackBone8x


class Radar_PillarRes18BackBone8x(PillarRes18BackBone8x):
    IN_PREFIX = "radar_"
