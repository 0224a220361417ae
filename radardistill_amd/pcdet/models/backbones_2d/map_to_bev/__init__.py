from .pointpillar_scatter import PointPillarScatter

# registry keyed by the yaml NAME (pcdet/models/backbones_2d/map_to_bev/__init__.py)
__all__ = {
    'PointPillarScatter': PointPillarScatter,
}
