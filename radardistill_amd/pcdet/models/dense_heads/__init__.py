from .center_head import CenterHead
from .radar_center_head import Radar_CenterHead

# registry keyed by the yaml NAME (pcdet/models/dense_heads/__init__.py:13-24); distill-config entries only
__all__ = {
    'CenterHead': CenterHead,
    'Radar_CenterHead': Radar_CenterHead,
}
