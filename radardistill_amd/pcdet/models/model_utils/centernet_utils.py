"""CenterNet geometry helpers behind the CenterHead (module path and public names of the reference's
pcdet/models/model_utils/centernet_utils.py; written from the semantics, arithmetic order kept where parity needs it).

Training targets are produced on the device by targets.hip (`rd_center_targets`); the host functions here serve the CPU / waymo
fallback of `assign_targets` and are what the GPU kernel is checked against (fixture g4 pins them: heat-maps bit-exact).
The eval-time decode works on rows x channels (channels-last) maps, the layout every dense kernel of this build produces.
"""
import numpy as np
import torch


# ------------------------------------------------------------------------------------------ gaussian targets (host)
def gaussian_radius(height, width, min_overlap=0.5):
    """Largest centre displacement r that keeps a (height x width) box at IoU >= min_overlap with itself, minimum over the three
    ways two such boxes can overlap (CornerNet).  Each case is the larger root of a r^2 - b r + c = 0; the fp32 operation order
    of every coefficient follows centernet_utils.py:9-35 so that int(radius) -- and with it every heat-map cell -- is bit-equal.
    height / width: fp32 tensors (box extent in cells)."""
    o = min_overlap
    span = height + width
    quadratics = (
        # (b, 4 a c)                                                                 a        c
        (span, 4 * (width * height * (1 - o) / (1 + o))),                          # 1        area (1-o)/(1+o)
        (2 * span, 16 * ((1 - o) * width * height)),                               # 4        (1-o) area
        (-2 * o * span, 4 * (4 * o) * ((o - 1) * width * height)),                 # 4 o      (o-1) area
    )
    roots = [(b + (b ** 2 - four_ac).sqrt()) / 2 for b, four_ac in quadratics]
    return torch.stack(roots, dim=0).amin(dim=0)


_PATCHES = {}


def gaussian_patch(radius):
    """fp32 (2r+1) x (2r+1) bump exp(-d^2 / 2 sigma^2) with sigma = (2r+1)/6, evaluated in float64 and cut below
    eps * peak before the cast (centernet_utils.py:38-44 as called from :50-51).  Cached per radius."""
    patch = _PATCHES.get(radius)
    if patch is None:
        side = 2 * radius + 1
        sigma = side / 6
        axis = np.arange(-radius, radius + 1, dtype=np.float64)
        bump = np.exp(-(axis[None, :] * axis[None, :] + axis[:, None] * axis[:, None]) / (2 * sigma * sigma))
        bump[bump < np.finfo(bump.dtype).eps * bump.max()] = 0
        patch = _PATCHES[radius] = torch.from_numpy(bump).float()
    return patch


def draw_gaussian_to_heatmap(heatmap, center, radius, k=1, valid_mask=None):
    """heatmap (H, W) <- max(heatmap, k * bump centred on the integer cell `center` = (x, y)), clipped at the map border."""
    H, W = heatmap.shape[0:2]
    cx, cy, r = int(center[0]), int(center[1]), int(radius)
    x0, x1 = max(cx - r, 0), min(cx + r + 1, W)
    y0, y1 = max(cy - r, 0), min(cy + r + 1, H)
    if x1 <= x0 or y1 <= y0:
        return heatmap
    bump = gaussian_patch(r)[y0 - (cy - r):y1 - (cy - r), x0 - (cx - r):x1 - (cx - r)].to(heatmap.device)
    if valid_mask is not None:
        bump = bump * valid_mask[y0:y1, x0:x1].float()
    window = heatmap[y0:y1, x0:x1]
    torch.maximum(window, bump * k, out=window)
    return heatmap


# ------------------------------------------------------------------------------------------ axis-aligned DIoU (IouRegLoss, torch path)
def bbox3d_overlaps_diou(pred_boxes, gt_boxes):
    """Distance-IoU of axis-aligned 3-D boxes (x, y, z, dx, dy, dz, ...): IoU - |c_p - c_g|^2 / diag(enclosing box)^2, in [-1, 1]
    (centernet_utils.py:462-497; the fused loss kernel centerloss.hip computes the same expression per object slot)."""
    if pred_boxes.shape[0] != gt_boxes.shape[0]:
        raise ValueError("bbox3d_overlaps_diou: the two box lists must pair up")
    p_lo, p_hi = pred_boxes[:, 0:3] - 0.5 * pred_boxes[:, 3:6], pred_boxes[:, 0:3] + 0.5 * pred_boxes[:, 3:6]
    g_lo, g_hi = gt_boxes[:, 0:3] - 0.5 * gt_boxes[:, 3:6], gt_boxes[:, 0:3] + 0.5 * gt_boxes[:, 3:6]
    common = torch.clamp(torch.minimum(p_hi, g_hi) - torch.maximum(p_lo, g_lo), min=0)
    hull = torch.clamp(torch.maximum(p_hi, g_hi) - torch.minimum(p_lo, g_lo), min=0)
    vol_common = common[:, 0] * common[:, 1] * common[:, 2]
    vol_pred = pred_boxes[:, 3] * pred_boxes[:, 4] * pred_boxes[:, 5]
    vol_gt = gt_boxes[:, 3] * gt_boxes[:, 4] * gt_boxes[:, 5]
    centre_dist2 = torch.pow(gt_boxes[:, 0:3] - pred_boxes[:, 0:3], 2).sum(-1)
    hull_diag2 = hull[:, 0] ** 2 + hull[:, 1] ** 2 + hull[:, 2] ** 2
    return torch.clamp(vol_common / (vol_gt + vol_pred - vol_common) - centre_dist2 / hull_diag2, min=-1.0, max=1.0)


# ------------------------------------------------------------------------------------------ inference decode
def _rows(x):
    """(B, C, H, W) -> (B, H*W, C); free for the channels-last maps the dense kernels write."""
    return x.permute(0, 2, 3, 1).reshape(x.shape[0], -1, x.shape[1])


def _peaks(scores, K):
    """The K highest entries of each sample's (C, H, W) score volume, descending: (score, cell = y*W + x, class, y, x).
    (The reference takes K per class and then K of those, centernet_utils.py:155-171 -- the same set, since every global top-K
    entry is inside its own class's top K.)"""
    B, C, H, W = scores.shape
    top, flat = torch.topk(_rows(scores).reshape(B, -1), K)              # index = cell * C + class
    cell = torch.div(flat, C, rounding_mode='floor')
    return top, cell, (flat - cell * C).int(), torch.div(cell, W, rounding_mode='floor').float(), (cell % W).float()


def _at(feature_map, cell):
    """Rows of a (B, C, H, W) map at `cell` (B, K) -> (B, K, C)."""
    rows = _rows(feature_map)
    return torch.take_along_dim(rows, cell.unsqueeze(-1).expand(-1, -1, rows.shape[-1]), dim=1)


def decode_bbox_from_heatmap(heatmap, rot_cos, rot_sin, center, center_z, dim, iou=None, rectifier=0.,
                             point_cloud_range=None, voxel_size=None, feature_map_stride=None, vel=None, K=100,
                             circle_nms=False, score_thresh=None, post_center_limit_range=None):
    """Boxes of the K strongest heat-map peaks per sample: centre = (peak cell + predicted sub-cell offset) * stride * voxel + range
    origin, z / size taken as predicted (`dim` already exponentiated by the caller), yaw = atan2(sin, cos), optional velocity;
    kept when the centre lies inside `post_center_limit_range` and the score exceeds `score_thresh`; with an IoU head the score
    becomes score^(1-rectifier) * clamp(iou, 0, 1)^rectifier (centernet_utils.py:231-308, the IoU-rectified variant of this fork).
    Returns one {'pred_boxes', 'pred_scores', 'pred_labels'} per sample (labels local to the head)."""
    if circle_nms:
        raise NotImplementedError("circle_nms is 'not checked yet' in the reference (centernet_utils.py:236-239)")
    if post_center_limit_range is None:
        raise ValueError("decode_bbox_from_heatmap needs post_center_limit_range")
    K = min(int(K), heatmap.shape[1] * heatmap.shape[2] * heatmap.shape[3])
    scores, cell, labels, ys, xs = _peaks(heatmap, K)
    offset = _at(center, cell)
    origin = offset.new_tensor([float(point_cloud_range[0]), float(point_cloud_range[1])])
    pitch = offset.new_tensor([float(voxel_size[0]), float(voxel_size[1])])
    xy = (torch.stack((xs, ys), dim=-1) + offset) * feature_map_stride * pitch + origin
    parts = [xy, _at(center_z, cell), _at(dim, cell), torch.atan2(_at(rot_sin, cell), _at(rot_cos, cell))]
    if vel is not None:
        parts.append(_at(vel, cell))
    boxes = torch.cat(parts, dim=-1)
    keep = ((boxes[..., :3] >= post_center_limit_range[:3]) & (boxes[..., :3] <= post_center_limit_range[3:])).all(-1)
    if score_thresh is not None:
        keep &= scores > score_thresh
    if iou is not None:
        quality = torch.clamp(_at(iou, cell).squeeze(-1), min=0, max=1.)
        scores = torch.pow(scores, 1 - rectifier) * torch.pow(quality, rectifier)
    return [{'pred_boxes': boxes[b, keep[b]], 'pred_scores': scores[b, keep[b]], 'pred_labels': labels[b, keep[b]]}
            for b in range(boxes.shape[0])]
