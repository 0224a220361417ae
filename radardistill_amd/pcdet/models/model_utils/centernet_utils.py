"""CenterNet target helpers with the reference's arithmetic (pcdet/models/model_utils/centernet_utils.py:9-69,462-497)."""
import numpy as np
import torch


def gaussian_radius(height, width, min_overlap=0.5):
    a1 = 1
    b1 = (height + width)
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    sq1 = (b1 ** 2 - 4 * a1 * c1).sqrt()
    r1 = (b1 + sq1) / 2
    a2 = 4
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    sq2 = (b2 ** 2 - 4 * a2 * c2).sqrt()
    r2 = (b2 + sq2) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    sq3 = (b3 ** 2 - 4 * a3 * c3).sqrt()
    r3 = (b3 + sq3) / 2
    return torch.min(torch.min(r1, r2), r3)


_GAUSS_CACHE = {}


def gaussian2D(shape, sigma=1):
    m, n = [(ss - 1.) / 2. for ss in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def gaussian_patch(radius):
    """float32 (2r+1, 2r+1) patch, sigma = (2r+1)/6, computed in float64 like the reference then cast."""
    g = _GAUSS_CACHE.get(radius)
    if g is None:
        d = 2 * radius + 1
        g = torch.from_numpy(gaussian2D((d, d), sigma=d / 6)).float()
        _GAUSS_CACHE[radius] = g
    return g


def draw_gaussian_to_heatmap(heatmap, center, radius, k=1, valid_mask=None):
    gaussian = gaussian_patch(int(radius))
    x, y = int(center[0]), int(center[1])
    height, width = heatmap.shape[0:2]
    left, right = min(x, radius), min(width - x, radius + 1)
    top, bottom = min(y, radius), min(height - y, radius + 1)
    masked_heatmap = heatmap[y - top:y + bottom, x - left:x + right]
    masked_gaussian = gaussian[radius - top:radius + bottom, radius - left:radius + right].to(heatmap.device)
    if min(masked_gaussian.shape) > 0 and min(masked_heatmap.shape) > 0:
        if valid_mask is not None:
            masked_gaussian = masked_gaussian * valid_mask[y - top:y + bottom, x - left:x + right].float()
        torch.max(masked_heatmap, masked_gaussian * k, out=masked_heatmap)
    return heatmap


def bbox3d_overlaps_diou(pred_boxes, gt_boxes):
    """Axis-aligned 3-D DIoU (centernet_utils.py:462-497)."""
    assert pred_boxes.shape[0] == gt_boxes.shape[0]
    qmin, qmax = pred_boxes[:, :2] - 0.5 * pred_boxes[:, 3:5], pred_boxes[:, :2] + 0.5 * pred_boxes[:, 3:5]
    gmin, gmax = gt_boxes[:, :2] - 0.5 * gt_boxes[:, 3:5], gt_boxes[:, :2] + 0.5 * gt_boxes[:, 3:5]
    inter_max_xy, inter_min_xy = torch.minimum(qmax, gmax), torch.maximum(qmin, gmin)
    out_max_xy, out_min_xy = torch.maximum(qmax, gmax), torch.minimum(qmin, gmin)
    volume_pred = pred_boxes[:, 3] * pred_boxes[:, 4] * pred_boxes[:, 5]
    volume_gt = gt_boxes[:, 3] * gt_boxes[:, 4] * gt_boxes[:, 5]
    inter_h = torch.minimum(pred_boxes[:, 2] + 0.5 * pred_boxes[:, 5], gt_boxes[:, 2] + 0.5 * gt_boxes[:, 5]) - \
        torch.maximum(pred_boxes[:, 2] - 0.5 * pred_boxes[:, 5], gt_boxes[:, 2] - 0.5 * gt_boxes[:, 5])
    inter_h = torch.clamp(inter_h, min=0)
    inter = torch.clamp(inter_max_xy - inter_min_xy, min=0)
    volume_inter = inter[:, 0] * inter[:, 1] * inter_h
    volume_union = volume_gt + volume_pred - volume_inter
    inter_diag = torch.pow(gt_boxes[:, 0:3] - pred_boxes[:, 0:3], 2).sum(-1)
    outer_h = torch.maximum(gt_boxes[:, 2] + 0.5 * gt_boxes[:, 5], pred_boxes[:, 2] + 0.5 * pred_boxes[:, 5]) - \
        torch.minimum(gt_boxes[:, 2] - 0.5 * gt_boxes[:, 5], pred_boxes[:, 2] - 0.5 * pred_boxes[:, 5])
    outer_h = torch.clamp(outer_h, min=0)
    outer = torch.clamp(out_max_xy - out_min_xy, min=0)
    outer_diag = outer[:, 0] ** 2 + outer[:, 1] ** 2 + outer_h ** 2
    return torch.clamp(volume_inter / volume_union - inter_diag / outer_diag, min=-1.0, max=1.0)


# ------------------------------------------------------------------------------------------ inference decode
def _gather_feat(feat, ind, mask=None):
    dim = feat.size(2)
    ind = ind.unsqueeze(2).expand(ind.size(0), ind.size(1), dim)
    feat = feat.gather(1, ind)
    if mask is not None:
        mask = mask.unsqueeze(2).expand_as(feat)
        feat = feat[mask].view(-1, dim)
    return feat


def _transpose_and_gather_feat(feat, ind):
    feat = feat.permute(0, 2, 3, 1).contiguous()
    feat = feat.view(feat.size(0), -1, feat.size(3))
    return _gather_feat(feat, ind)


def _topk(scores, K=40):
    """Top-K peaks over classes and cells (centernet_utils.py:155-171)."""
    batch, num_class, height, width = scores.size()
    topk_scores, topk_inds = torch.topk(scores.flatten(2, 3), K)
    topk_inds = topk_inds % (height * width)
    topk_ys = (topk_inds // width).float()
    topk_xs = (topk_inds % width).int().float()
    topk_score, topk_ind = torch.topk(topk_scores.view(batch, -1), K)
    topk_classes = (topk_ind // K).int()
    topk_inds = _gather_feat(topk_inds.view(batch, -1, 1), topk_ind).view(batch, K)
    topk_ys = _gather_feat(topk_ys.view(batch, -1, 1), topk_ind).view(batch, K)
    topk_xs = _gather_feat(topk_xs.view(batch, -1, 1), topk_ind).view(batch, K)
    return topk_score, topk_inds, topk_classes, topk_ys, topk_xs


def decode_bbox_from_heatmap(heatmap, rot_cos, rot_sin, center, center_z, dim, iou=None, rectifier=0.,
                             point_cloud_range=None, voxel_size=None, feature_map_stride=None, vel=None, K=100,
                             circle_nms=False, score_thresh=None, post_center_limit_range=None):
    """centernet_utils.py:231-308 (the IoU-rectified variant this fork uses).  Dense maps may be channels-last views; only K
    cells per sample are gathered from them."""
    batch_size, num_class, _, _ = heatmap.size()
    if circle_nms:
        raise NotImplementedError("circle_nms is 'not checked yet' in the reference (centernet_utils.py:236-239)")
    scores, inds, class_ids, ys, xs = _topk(heatmap, K=K)
    center = _transpose_and_gather_feat(center, inds).view(batch_size, K, 2)
    rot_sin = _transpose_and_gather_feat(rot_sin, inds).view(batch_size, K, 1)
    rot_cos = _transpose_and_gather_feat(rot_cos, inds).view(batch_size, K, 1)
    center_z = _transpose_and_gather_feat(center_z, inds).view(batch_size, K, 1)
    dim = _transpose_and_gather_feat(dim, inds).view(batch_size, K, 3)
    if iou is not None:
        iou = _transpose_and_gather_feat(iou, inds).view(batch_size, K, 1)
    angle = torch.atan2(rot_sin, rot_cos)
    xs = xs.view(batch_size, K, 1) + center[:, :, 0:1]
    ys = ys.view(batch_size, K, 1) + center[:, :, 1:2]
    xs = xs * feature_map_stride * voxel_size[0] + point_cloud_range[0]
    ys = ys * feature_map_stride * voxel_size[1] + point_cloud_range[1]
    box_part_list = [xs, ys, center_z, dim, angle]
    if vel is not None:
        vel = _transpose_and_gather_feat(vel, inds).view(batch_size, K, 2)
        box_part_list.append(vel)
    final_box_preds = torch.cat(box_part_list, dim=-1)
    final_scores = scores.view(batch_size, K)
    final_class_ids = class_ids.view(batch_size, K)
    assert post_center_limit_range is not None
    mask = (final_box_preds[..., :3] >= post_center_limit_range[:3]).all(2)
    mask &= (final_box_preds[..., :3] <= post_center_limit_range[3:]).all(2)
    if score_thresh is not None:
        mask &= (final_scores > score_thresh)
    ret_pred_dicts = []
    for k in range(batch_size):
        cur_mask = mask[k]
        cur_boxes = final_box_preds[k, cur_mask]
        cur_scores = final_scores[k, cur_mask]
        cur_labels = final_class_ids[k, cur_mask]
        if iou is not None:
            iou_preds = torch.clamp(iou[k, cur_mask].view(-1), min=0, max=1.)
            cur_scores = torch.pow(cur_scores, 1 - rectifier) * torch.pow(iou_preds, rectifier)
        ret_pred_dicts.append({'pred_boxes': cur_boxes, 'pred_scores': cur_scores, 'pred_labels': cur_labels})
    return ret_pred_dicts
