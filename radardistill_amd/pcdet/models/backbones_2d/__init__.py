from .base_bev_backbone import BaseBEVBackboneV2
from .radar_distill_final import Radar_Distill

# registry keyed by the yaml NAME (pcdet/models/backbones_2d/__init__.py:8-14); distill-config entries only
__all__ = {
    'BaseBEVBackboneV2': BaseBEVBackboneV2,
    'Radar_Distill': Radar_Distill,
}
