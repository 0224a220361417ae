"""Oracle: the whole RadarDistill training forward (PillarNet with DISTILL=True) as one function
(SURVEY 8(a) row A12).  Test infrastructure.

Follows pcdet/models/detectors/pillarnet.py:12-73 (freeze list, module chain, get_training_distll_loss)
and the module order of pcdet/models/detectors/detector3d_template.py:23-26.
"""
import torch

from . import bev, head, sparse, vfe

CLASS_NAMES = ["car", "truck", "construction_vehicle", "bus", "trailer",
               "barrier", "motorcycle", "bicycle", "pedestrian", "traffic_cone"]
HEADS = [["car"], ["truck", "construction_vehicle"], ["bus", "trailer"], ["barrier"],
         ["motorcycle", "bicycle"], ["pedestrian", "traffic_cone"]]


def forward_train(state, batch, pc_range, voxel_size, grid_size, run_teacher_head=False):
    """state: dict name -> tensor with the reference's state_dict names (teacher unprefixed, student
    `radar_` prefixed).  Trainable student tensors may have requires_grad=True.
    batch: dict(points (N,6), radar_points (M,7), gt_boxes (B,K,10), batch_size).
    Teacher modules run in eval mode (pillarnet.py:31-33), student in train mode.
    Returns (loss, tb_dict, intermediates)."""
    B = int(batch["batch_size"])
    inter = {}
    with torch.no_grad():
        tv = vfe.dynamic_pillar_vfe(batch["points"], state, "vfe.", pc_range, voxel_size, grid_size, training=False)
        tb3 = sparse.pillar_res18_backbone(tv["pillar_features"], tv["pillar_coords"].numpy(), B, grid_size,
                                           state, "backbone_3d.", training=False)
        t_up, t_feat = bev.dense_enc(tb3["x_conv4"], tb3["x_conv5"], state, "backbone_2d.", training=False)
        if run_teacher_head:
            inter["lidar_pred_dicts"] = head.center_head_forward(t_feat, state, "dense_head.", len(HEADS), False)
    rv = vfe.dynamic_pillar_vfe(batch["radar_points"], state, "radar_vfe.", pc_range, voxel_size, grid_size, training=True)
    rb3 = sparse.pillar_res18_backbone(rv["pillar_features"], rv["pillar_coords"].numpy(), B, grid_size,
                                       state, "radar_backbone_3d.", training=True)
    r2d = bev.radar_distill_forward(rb3["x_conv4"], rb3["x_conv5"], state, "radar_backbone_2d.", training=True)
    preds = head.center_head_forward(r2d["radar_spatial_features_2d"], state, "radar_dense_head.", len(HEADS), True)
    fmap_hw = r2d["radar_spatial_features_2d"].shape[2:]
    targets = head.assign_targets(batch["gt_boxes"], fmap_hw, CLASS_NAMES, HEADS, pc_range, voxel_size)
    loss_feat, tb = bev.distill_loss(tb3["x_conv4"], r2d, t_feat, t_up, targets["heatmaps"], [p["hm"] for p in preds])
    loss_rpn, tb2 = head.center_head_loss(preds, targets, voxel_size, pc_range)
    tb.update(tb2)
    inter.update(dict(teacher_vfe=tv, teacher_b3=tb3, teacher_up=t_up, teacher_feat=t_feat, radar_vfe=rv,
                      radar_b3=rb3, radar_2d=r2d, preds=preds, targets=targets,
                      loss_feature=loss_feat, loss_rpn=loss_rpn))
    return loss_feat + loss_rpn, tb, inter


def forward_radar_only(state, radar_points, batch_size, pc_range, voxel_size, grid_size):
    """BASELINE configs[0] (C1): the radar student alone in eval mode -- radar VFE -> SparseEnc(+conv5) -> CMA + DenseEnc -> CenterHead
    forward (the module chain radar_distill_val.yaml builds: detectors/pillarnet.py:28-46 over the four radar_* modules)."""
    rv = vfe.dynamic_pillar_vfe(radar_points, state, "radar_vfe.", pc_range, voxel_size, grid_size, training=False)
    rb3 = sparse.pillar_res18_backbone(rv["pillar_features"], rv["pillar_coords"].numpy(), batch_size, grid_size, state, "radar_backbone_3d.",
                                       training=False)
    r2d = bev.radar_distill_forward(rb3["x_conv4"], rb3["x_conv5"], state, "radar_backbone_2d.", training=False)
    return head.center_head_forward(r2d["radar_spatial_features_2d"], state, "radar_dense_head.", len(HEADS), False)
