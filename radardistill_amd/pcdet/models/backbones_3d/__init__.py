from .spconv_backbone_2d import PillarRes18BackBone8x
from .spconv_backbone_2d_distillation import Radar_PillarRes18BackBone8x

# registry keyed by the yaml NAME (pcdet/models/backbones_3d/__init__.py:12-27); distill-config entries only
__all__ = {
    'PillarRes18BackBone8x': PillarRes18BackBone8x,
    'Radar_PillarRes18BackBone8x': Radar_PillarRes18BackBone8x,
}
