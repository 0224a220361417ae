"""Teacher CenterHead (pcdet/models/dense_heads/center_head.py:48-424): same network as the student head, reads
`spatial_features_2d`, and with DISTILL_PRED only publishes `lidar_pred_dicts` (:403-406)."""
from .radar_center_head import Radar_CenterHead, SeparateHead  # noqa: F401


class CenterHead(Radar_CenterHead):
    FEATURE_KEY = 'spatial_features_2d'
    IS_TEACHER = True
