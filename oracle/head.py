"""Oracle: CenterHead forward, target assignment and detection losses.
Test infrastructure.

Follows
  pcdet/models/dense_heads/radar_center_head.py:28-62    (SeparateHead)
  pcdet/models/dense_heads/radar_center_head.py:128-252  (assign_target_of_single_head, assign_targets)
  pcdet/models/dense_heads/radar_center_head.py:258-330  (get_loss)
  pcdet/models/dense_heads/radar_center_head.py:409-440  (forward)
  pcdet/utils/loss_utils.py:266-301 (neg_loss_cornernet), :347-376 (_reg_loss), :390-394
      (_transpose_and_gather_feat), :651-673 (IouLoss), :677-701 (IouRegLoss)
  pcdet/models/model_utils/centernet_utils.py:9-35 (gaussian_radius), :38-69 (gaussian2D, draw),
      :462-497 (bbox3d_overlaps_diou)
  pcdet/ops/iou3d_nms/iou3d_nms_utils.py:83-117 (boxes_aligned_iou3d_gpu)
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn.functional as F

HEAD_ORDER = ("center", "center_z", "dim", "rot", "vel", "iou")
HEAD_OUT = {"center": 2, "center_z": 1, "dim": 3, "rot": 2, "vel": 2, "iou": 1}


# ----------------------------------------------------------------------------- forward
def _bn(x, state, prefix, training):
    return F.batch_norm(x, state[prefix + "running_mean"], state[prefix + "running_var"],
                        state[prefix + "weight"], state[prefix + "bias"], training=training,
                        momentum=0.1, eps=1e-5)


def center_head_forward(feat, state, prefix, n_heads, training):
    """shared_conv (Conv3x3 256->64 bias + BN + ReLU) then per task head 7 branches of
    [Conv3x3(64->64, bias)+BN+ReLU, Conv3x3(64->out, bias)] (radar_center_head.py:88-114, 28-62)."""
    x = F.conv2d(feat, state[prefix + "shared_conv.0.weight"], state[prefix + "shared_conv.0.bias"], padding=1)
    x = F.relu(_bn(x, state, prefix + "shared_conv.1.", training))
    preds = []
    for h in range(n_heads):
        d = {}
        for name in HEAD_ORDER + ("hm",):
            p = f"{prefix}heads_list.{h}.{name}."
            y = F.conv2d(x, state[p + "0.0.weight"], state[p + "0.0.bias"], padding=1)
            y = F.relu(_bn(y, state, p + "0.1.", training))
            d[name] = F.conv2d(y, state[p + "1.weight"], state[p + "1.bias"], padding=1)
        preds.append(d)
    return preds


# ----------------------------------------------------------------------------- targets
def gaussian_radius(height, width, min_overlap=0.5):
    """centernet_utils.py:9-35 (torch, fp32)."""
    a1 = 1
    b1 = height + width
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + (b1 ** 2 - 4 * a1 * c1).sqrt()) / 2
    a2 = 4
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    r2 = (b2 + (b2 ** 2 - 4 * a2 * c2).sqrt()) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    r3 = (b3 + (b3 ** 2 - 4 * a3 * c3).sqrt()) / 2
    return torch.min(torch.min(r1, r2), r3)


def gaussian2d(radius):
    """centernet_utils.py:38-44 with shape (2r+1, 2r+1), sigma = (2r+1)/6: float64 numpy."""
    d = 2 * radius + 1
    sigma = d / 6
    m = (d - 1.0) / 2.0
    y, x = np.ogrid[-m:m + 1, -m:m + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def draw_gaussian(heatmap, center, radius):
    """centernet_utils.py:47-69 (k=1, no valid_mask): elementwise max of the fp32-cast gaussian."""
    g = gaussian2d(radius)
    x, y = int(center[0]), int(center[1])
    H, W = heatmap.shape
    left, right = min(x, radius), min(W - x, radius + 1)
    top, bottom = min(y, radius), min(H - y, radius + 1)
    mh = heatmap[y - top:y + bottom, x - left:x + right]
    mg = torch.from_numpy(g[radius - top:radius + bottom, radius - left:radius + right]).float()
    if min(mg.shape) > 0 and min(mh.shape) > 0:
        torch.max(mh, mg, out=mh)


def assign_single_head(num_classes, gt_boxes, fmap_xy, stride, pc_range, voxel_size,
                       num_max_objs=500, overlap=0.1, min_radius=2):
    """radar_center_head.py:128-187.  gt_boxes (n, 10) with the last column = 1-based class id
    within this head.  fmap_xy = [x, y]."""
    fx, fy = int(fmap_xy[0]), int(fmap_xy[1])
    heatmap = gt_boxes.new_zeros(num_classes, fy, fx)
    ret_boxes = gt_boxes.new_zeros((num_max_objs, gt_boxes.shape[-1]))
    gt_box = gt_boxes.new_zeros((num_max_objs, gt_boxes.shape[-1] - 3))
    inds = gt_boxes.new_zeros(num_max_objs).long()
    mask = gt_boxes.new_zeros(num_max_objs).long()
    x, y, z = gt_boxes[:, 0], gt_boxes[:, 1], gt_boxes[:, 2]
    cx = (x - pc_range[0]) / voxel_size[0] / stride
    cy = (y - pc_range[1]) / voxel_size[1] / stride
    cx = torch.clamp(cx, min=0, max=fx - 0.5)
    cy = torch.clamp(cy, min=0, max=fy - 0.5)
    center = torch.cat((cx[:, None], cy[:, None]), dim=-1)
    center_int = center.int()
    dx = gt_boxes[:, 3] / voxel_size[0] / stride
    dy = gt_boxes[:, 4] / voxel_size[1] / stride
    radius = torch.clamp_min(gaussian_radius(dx, dy, min_overlap=overlap).int(), min=min_radius)
    for k in range(min(num_max_objs, gt_boxes.shape[0])):
        if dx[k] <= 0 or dy[k] <= 0:
            continue
        if not (0 <= center_int[k][0] <= fx and 0 <= center_int[k][1] <= fy):
            continue
        cls = (gt_boxes[k, -1] - 1).long()
        draw_gaussian(heatmap[cls], center[k], int(radius[k].item()))
        inds[k] = center_int[k, 1] * fx + center_int[k, 0]
        mask[k] = 1
        ret_boxes[k, 0:2] = center[k] - center_int[k].float()
        ret_boxes[k, 2] = z[k]
        ret_boxes[k, 3:6] = gt_boxes[k, 3:6].log()
        ret_boxes[k, 6] = torch.cos(gt_boxes[k, 6])
        ret_boxes[k, 7] = torch.sin(gt_boxes[k, 6])
        ret_boxes[k, 8:] = gt_boxes[k, 7:-1]
        gt_box[k, :7] = gt_boxes[k, :7]
    return heatmap, ret_boxes, inds, mask, gt_box


def assign_targets(gt_boxes, fmap_hw, class_names, class_names_each_head, pc_range, voxel_size,
                   stride=8, num_max_objs=500, overlap=0.1, min_radius=2):
    """radar_center_head.py:189-252.  gt_boxes (B, M, 10): [x,y,z,dx,dy,dz,yaw,vx,vy,class(1-based, 0 = pad)].

    NOTE the reference rewrites the class column of `gt_boxes` IN PLACE (`temp_box[-1] = ...`, :222):
    `temp_box` is a view.  Within one call that is harmless because rewritten ids (1 or 2) name
    classes of heads that were already processed; this restatement works on a clone and therefore
    leaves the caller's tensor untouched (the teacher head runs in eval mode and never assigns,
    pillarnet.py:31-33 + center_head.py:397, so there is no second call on the mutated tensor)."""
    gt_boxes = gt_boxes.clone()
    fmap_xy = list(fmap_hw)[::-1]
    B = gt_boxes.shape[0]
    all_names = np.array(["bg", *class_names])
    ret = {"heatmaps": [], "target_boxes": [], "inds": [], "masks": [], "gt_box": []}
    for head_names in class_names_each_head:
        hm_l, tb_l, ind_l, m_l, gb_l = [], [], [], [], []
        for b in range(B):
            cur = gt_boxes[b]
            names = all_names[cur[:, -1].long().numpy()]
            sel = []
            for i, name in enumerate(names):
                if name not in head_names:
                    continue
                tmp = cur[i]
                tmp[-1] = head_names.index(name) + 1
                sel.append(tmp[None, :])
            sel = torch.cat(sel, dim=0) if sel else cur[:0, :]
            hm, tb, ind, m, gb = assign_single_head(len(head_names), sel, fmap_xy, stride, pc_range, voxel_size,
                                                    num_max_objs, overlap, min_radius)
            hm_l.append(hm); tb_l.append(tb); ind_l.append(ind); m_l.append(m); gb_l.append(gb)
        ret["heatmaps"].append(torch.stack(hm_l)); ret["target_boxes"].append(torch.stack(tb_l))
        ret["inds"].append(torch.stack(ind_l)); ret["masks"].append(torch.stack(m_l))
        ret["gt_box"].append(torch.stack(gb_l))
    return ret


# ----------------------------------------------------------------------------- losses
def focal_loss(pred, gt):
    """neg_loss_cornernet (loss_utils.py:266-301), mask=None."""
    pos = gt.eq(1).float()
    neg = gt.lt(1).float()
    neg_w = torch.pow(1 - gt, 4)
    pos_loss = (torch.log(pred) * torch.pow(1 - pred, 2) * pos).sum()
    neg_loss = (torch.log(1 - pred) * torch.pow(pred, 2) * neg_w * neg).sum()
    num_pos = pos.sum()
    if num_pos == 0:
        return -neg_loss
    return -(pos_loss + neg_loss) / num_pos


def gather_feat(feat, ind):
    """_transpose_and_gather_feat (loss_utils.py:379-394): (B,C,H,W),(B,K) -> (B,K,C)."""
    B, C = feat.shape[:2]
    f = feat.permute(0, 2, 3, 1).reshape(B, -1, C)
    return f.gather(1, ind.unsqueeze(2).expand(B, ind.shape[1], C))


def reg_loss(pred_map, mask, ind, target):
    """RegLossCenterNet / _reg_loss (loss_utils.py:347-376, 397-419): per-channel masked L1 / max(num,1)."""
    pred = gather_feat(pred_map, ind)
    num = mask.float().sum()
    m = mask.unsqueeze(2).expand_as(target).float() * (~torch.isnan(target)).float()
    loss = torch.abs(pred * m - target * m).transpose(2, 0).sum(dim=2).sum(dim=1)
    return loss / torch.clamp_min(num, min=1.0)


_LIB = None


def _iou_lib():
    global _LIB
    if _LIB is None:
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.path.join(here, "_build", "liboracle_iou3d.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        _LIB = ctypes.CDLL(path)
        _LIB.oracle_boxes_aligned_overlap_bev.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return _LIB


def boxes_aligned_overlap_bev(a, b):
    """iou3d_nms_kernel.cu:266-277 via the C restatement oracle/iou3d.c."""
    a = np.ascontiguousarray(a.detach().numpy(), dtype=np.float32)
    b = np.ascontiguousarray(b.detach().numpy(), dtype=np.float32)
    out = np.zeros(a.shape[0], dtype=np.float32)
    if a.shape[0]:
        _iou_lib().oracle_boxes_aligned_overlap_bev(a.shape[0], a.ctypes.data, b.ctypes.data, out.ctypes.data)
    return torch.from_numpy(out)


def boxes_aligned_iou3d(a, b):
    """boxes_aligned_iou3d_gpu (iou3d_nms_utils.py:83-117) -> (N,1)."""
    a_max = (a[:, 2] + a[:, 5] / 2).view(-1, 1); a_min = (a[:, 2] - a[:, 5] / 2).view(-1, 1)
    b_max = (b[:, 2] + b[:, 5] / 2).view(-1, 1); b_min = (b[:, 2] - b[:, 5] / 2).view(-1, 1)
    bev = boxes_aligned_overlap_bev(a, b).view(-1, 1)
    oh = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    o3d = bev * oh
    va = (a[:, 3] * a[:, 4] * a[:, 5]).view(-1, 1)
    vb = (b[:, 3] * b[:, 4] * b[:, 5]).view(-1, 1)
    return o3d / torch.clamp(va + vb - o3d, min=1e-6)


def diou(p, g):
    """bbox3d_overlaps_diou (centernet_utils.py:462-497): axis-aligned 3-D DIoU."""
    def corners(c, d):
        return c - 0.5 * d, c + 0.5 * d
    qmin, qmax = corners(p[:, :2], p[:, 3:5])
    gmin, gmax = corners(g[:, :2], g[:, 3:5])
    imax, imin = torch.minimum(qmax, gmax), torch.maximum(qmin, gmin)
    omax, omin = torch.maximum(qmax, gmax), torch.minimum(qmin, gmin)
    vp = p[:, 3] * p[:, 4] * p[:, 5]
    vg = g[:, 3] * g[:, 4] * g[:, 5]
    ih = torch.clamp(torch.minimum(p[:, 2] + 0.5 * p[:, 5], g[:, 2] + 0.5 * g[:, 5]) -
                     torch.maximum(p[:, 2] - 0.5 * p[:, 5], g[:, 2] - 0.5 * g[:, 5]), min=0)
    inter = torch.clamp(imax - imin, min=0)
    vi = inter[:, 0] * inter[:, 1] * ih
    vu = vg + vp - vi
    idiag = torch.pow(g[:, 0:3] - p[:, 0:3], 2).sum(-1)
    oh = torch.clamp(torch.maximum(g[:, 2] + 0.5 * g[:, 5], p[:, 2] + 0.5 * p[:, 5]) -
                     torch.minimum(g[:, 2] - 0.5 * g[:, 5], p[:, 2] - 0.5 * p[:, 5]), min=0)
    outer = torch.clamp(omax - omin, min=0)
    odiag = outer[:, 0] ** 2 + outer[:, 1] ** 2 + oh ** 2
    return torch.clamp(vi / vu - idiag / odiag, min=-1.0, max=1.0)


def decode_boxes(pred, stride, voxel_size, pc_range):
    """radar_center_head.py:285-312: every cell -> (x,y,z,dx,dy,dz,yaw) map (B,7,H,W).
    PARITY TRAP reproduced: `+ int(self.point_cloud_range[0])` truncates the range origin (:309-310)."""
    dim = torch.exp(torch.clamp(pred["dim"], min=-5, max=5))
    rot = torch.atan2(pred["rot"][:, 1:2], pred["rot"][:, 0:1])
    B, _, H, W = dim.shape
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    xs = xs.view(1, 1, H, W).to(dim) + pred["center"][:, 0:1]
    ys = ys.view(1, 1, H, W).to(dim) + pred["center"][:, 1:2]
    xs = xs * int(stride) * voxel_size[0] + int(pc_range[0])
    ys = ys * int(stride) * voxel_size[1] + int(pc_range[1])
    return torch.cat([xs, ys, pred["center_z"], dim, rot], dim=1)


def center_head_loss(preds, targets, voxel_size, pc_range, stride=8, cls_weight=1.0, loc_weight=0.25,
                     code_weights=(1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2, 1.0, 1.0)):
    """Radar_CenterHead.get_loss (radar_center_head.py:258-330) with with_iou and IOU_REG == DIoU."""
    loss = 0
    tb = {}
    cw = torch.tensor(code_weights, dtype=torch.float32)
    for idx, pred in enumerate(preds):
        hm = torch.clamp(torch.sigmoid(pred["hm"]), min=1e-4, max=1 - 1e-4)
        hm_loss = focal_loss(hm, targets["heatmaps"][idx]) * cls_weight
        pred_boxes = torch.cat([pred[n] for n in HEAD_ORDER], dim=1)[:, :-1]
        mask, ind = targets["masks"][idx], targets["inds"][idx]
        r = reg_loss(pred_boxes, mask, ind, targets["target_boxes"][idx])
        loc_loss = (r * cw).sum() * loc_weight
        loss = loss + hm_loss + loc_loss
        tb[f"hm_loss_head_{idx}"] = hm_loss
        tb[f"loc_loss_head_{idx}"] = loc_loss
        box_map = decode_boxes(pred, stride, voxel_size, pc_range)
        gt_box = targets["gt_box"][idx]
        mb = mask.bool()
        # IouLoss (loss_utils.py:651-673)
        if mask.sum() == 0:
            iou_loss = pred["iou"].new_zeros((1))
        else:
            p = gather_feat(pred["iou"], ind)[mb]
            pb = gather_feat(box_map.detach(), ind)
            tgt = 2 * boxes_aligned_iou3d(pb[mb], gt_box[mb]) - 1
            iou_loss = F.l1_loss(p, tgt, reduction="sum") / (mask.sum() + 1e-4)
        loss = loss + iou_loss
        tb[f"iou_loss_head_{idx}"] = iou_loss
        # IouRegLoss (loss_utils.py:677-701)
        if mask.sum() == 0:
            ireg = box_map.new_zeros((1))
        else:
            pb = gather_feat(box_map, ind)
            ireg = (1.0 - diou(pb[mb], gt_box[mb])).sum() / (mask.sum() + 1e-4)
        loss = loss + loc_weight * ireg
        tb[f"iou_reg_loss_head_{idx}"] = ireg
    tb["rpn_loss"] = loss
    return loss, tb
