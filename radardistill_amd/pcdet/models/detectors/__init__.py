from .detector3d_template import Detector3DTemplate
from .pillarnet import PillarNet

# registry keyed by MODEL.NAME (pcdet/models/detectors/__init__.py:19-38); the distill config uses PillarNet
__all__ = {
    'Detector3DTemplate': Detector3DTemplate,
    'PillarNet': PillarNet,
}


def build_detector(model_cfg, num_class, dataset):
    return __all__[model_cfg.NAME](model_cfg=model_cfg, num_class=num_class, dataset=dataset)
