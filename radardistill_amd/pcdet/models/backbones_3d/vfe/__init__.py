from .dynamic_pillar_vfe import DynamicPillarVFESimple2D, Radar_DynamicPillarVFESimple2D, Radar_DynamicPillarVFESimple2D_Test
from .pillar_vfe import PillarVFE
from .vfe_template import VFETemplate

# registry keyed by the yaml NAME (pcdet/models/backbones_3d/vfe/__init__.py:9-21); only the distill-config entries
__all__ = {
    'VFETemplate': VFETemplate,
    'DynamicPillarVFESimple2D': DynamicPillarVFESimple2D,
    'Radar_DynamicPillarVFESimple2D': Radar_DynamicPillarVFESimple2D,
    'Radar_DynamicPillarVFESimple2D_Test': Radar_DynamicPillarVFESimple2D_Test,      # radar_distill_val.yaml:67
    'PillarVFE': PillarVFE,                 # padded-voxel input format
}
