"""CPU oracle of the inference post-processing.  TEST INFRASTRUCTURE, NOT PRODUCT.

Restates, in loops / plain torch on the CPU:
  * nms_gpu            pcdet/ops/iou3d_nms/iou3d_nms_utils.py:119-137, src/iou3d_nms.cpp:137-183, iou3d_nms_kernel.cu:295-340
  * boxes_iou3d_gpu    pcdet/ops/iou3d_nms/iou3d_nms_utils.py:55-81
  * class_agnostic_nms pcdet/models/model_utils/model_nms_utils.py:6-25
  * _topk / decode_bbox_from_heatmap   pcdet/models/model_utils/centernet_utils.py:155-171, 231-308
  * generate_predicted_boxes           pcdet/models/dense_heads/radar_center_head.py:332-389
Pinning: decode + class_agnostic_nms against tests/golden/g6_decode.npz produced by the reference's own functions (with the absent
iou3d extension bridged by the C restatement here, so the rotated-overlap arithmetic itself stays "parity unpinned" vs the binary).
"""
import ctypes

import numpy as np
import torch

from . import head as ohead


def _lib():
    lib = ohead._iou_lib()
    lib.oracle_boxes_overlap_bev.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.oracle_nms_bev.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p]
    lib.oracle_nms_bev.restype = ctypes.c_int
    return lib


def nms_bev_sorted(boxes_sorted, thresh):
    """boxes (n,7) sorted by descending score -> kept indices (ascending), int64."""
    b = np.ascontiguousarray(boxes_sorted.detach().numpy(), dtype=np.float32)
    keep = np.zeros(max(b.shape[0], 1), dtype=np.int64)
    n = _lib().oracle_nms_bev(b.shape[0], b.ctypes.data, float(thresh), keep.ctypes.data) if b.shape[0] else 0
    return torch.from_numpy(keep[:n].copy())


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep = nms_bev_sorted(boxes[order][:, :7].contiguous(), thresh)
    return order[keep].contiguous(), None


def boxes_overlap_bev(a, b):
    a = np.ascontiguousarray(a.detach().numpy(), dtype=np.float32)
    b = np.ascontiguousarray(b.detach().numpy(), dtype=np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    if a.shape[0] and b.shape[0]:
        _lib().oracle_boxes_overlap_bev(a.shape[0], a.ctypes.data, b.shape[0], b.ctypes.data, out.ctypes.data)
    return torch.from_numpy(out)


def boxes_iou3d(a, b):
    a_max = (a[:, 2] + a[:, 5] / 2).view(-1, 1); a_min = (a[:, 2] - a[:, 5] / 2).view(-1, 1)
    b_max = (b[:, 2] + b[:, 5] / 2).view(1, -1); b_min = (b[:, 2] - b[:, 5] / 2).view(1, -1)
    bev = boxes_overlap_bev(a, b)
    oh = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    o3d = bev * oh
    va = (a[:, 3] * a[:, 4] * a[:, 5]).view(-1, 1)
    vb = (b[:, 3] * b[:, 4] * b[:, 5]).view(1, -1)
    return o3d / torch.clamp(va + vb - o3d, min=1e-6)


def class_agnostic_nms(box_scores, box_preds, nms_thresh, pre_maxsize, post_maxsize):
    if box_scores.shape[0] == 0:
        return torch.zeros(0, dtype=torch.long), box_scores[:0]
    k = min(pre_maxsize, box_scores.shape[0])
    scores_nms, indices = torch.topk(box_scores, k=k)
    keep, _ = nms_gpu(box_preds[indices][:, :7], scores_nms, nms_thresh)
    selected = indices[keep[:post_maxsize]]
    return selected, box_scores[selected]


def decode_sample(hm, center, center_z, dim, rot, vel, iou, K, stride, voxel_size, pc_range, score_thresh, limit_range, rectifier=0.0):
    """One sample of one head, loops over candidates.  hm (nc,H,W) AFTER sigmoid, dim AFTER exp, iou AFTER (x+1)/2 (or None);
    -> boxes (m, 7 or 9), scores (m,), local labels (m,) in the candidate order of the reference (_topk)."""
    nc, H, W = hm.shape
    flat = hm.reshape(nc, -1)
    k1 = min(K, flat.shape[1])
    s1, i1 = torch.topk(flat, k1)                                   # per class
    s2, i2 = torch.topk(s1.reshape(-1), K)                          # over classes
    cls = (i2 // k1).int()
    cell = i1.reshape(-1)[i2]
    boxes, scores, labels = [], [], []
    for j in range(K):
        c = int(cell[j]); y, x = c // W, c % W
        bx = (float(x) + center[0, y, x]) * stride * voxel_size[0] + pc_range[0]
        by = (float(y) + center[1, y, x]) * stride * voxel_size[1] + pc_range[1]
        bz = center_z[0, y, x]
        ang = torch.atan2(rot[1, y, x], rot[0, y, x])
        parts = [bx, by, bz, dim[0, y, x], dim[1, y, x], dim[2, y, x], ang]
        if vel is not None:
            parts += [vel[0, y, x], vel[1, y, x]]
        box = torch.stack([torch.as_tensor(p, dtype=torch.float32) for p in parts])
        ok = bool((box[:3] >= limit_range[:3]).all() and (box[:3] <= limit_range[3:]).all())
        sc = s2[j]
        if score_thresh is not None:
            ok = ok and bool(sc > score_thresh)
        if not ok:
            continue
        if iou is not None:
            q = torch.clamp(iou[y, x], min=0, max=1.0)
            sc = torch.pow(sc, 1 - rectifier) * torch.pow(q, rectifier)
        boxes.append(box); scores.append(sc); labels.append(int(cls[j]))
    nb = 9 if vel is not None else 7
    if not boxes:
        return torch.zeros((0, nb)), torch.zeros(0), torch.zeros(0, dtype=torch.long)
    return torch.stack(boxes), torch.stack(scores), torch.tensor(labels, dtype=torch.long)


def generate_predicted_boxes(pred_dicts, class_id_mapping_each_head, post_cfg, stride, voxel_size, pc_range, rectifier=0.0):
    """pred_dicts: per head {name: (B,c,H,W) raw network outputs} -> per sample dict of boxes / scores / labels (global, 1-based)."""
    B = pred_dicts[0]['hm'].shape[0]
    limit = torch.tensor(post_cfg['POST_CENTER_LIMIT_RANGE']).float()
    nms = post_cfg['NMS_CONFIG']
    out = [{'pred_boxes': [], 'pred_scores': [], 'pred_labels': []} for _ in range(B)]
    for h, pd in enumerate(pred_dicts):
        hm = pd['hm'].sigmoid(); dim = pd['dim'].exp()
        iou = (pd['iou'].squeeze(1) + 1) * 0.5 if 'iou' in pd else None
        for b in range(B):
            boxes, scores, labels = decode_sample(hm[b], pd['center'][b], pd['center_z'][b], dim[b], pd['rot'][b], pd.get('vel', [None] * B)[b],
                                                  None if iou is None else iou[b], post_cfg['MAX_OBJ_PER_SAMPLE'], stride, voxel_size, pc_range,
                                                  post_cfg['SCORE_THRESH'], limit, rectifier)
            labels = class_id_mapping_each_head[h][labels]
            sel, sel_scores = class_agnostic_nms(scores, boxes, nms['NMS_THRESH'], nms['NMS_PRE_MAXSIZE'], nms['NMS_POST_MAXSIZE'])
            out[b]['pred_boxes'].append(boxes[sel]); out[b]['pred_scores'].append(sel_scores); out[b]['pred_labels'].append(labels[sel])
    for b in range(B):
        out[b]['pred_boxes'] = torch.cat(out[b]['pred_boxes'], 0)
        out[b]['pred_scores'] = torch.cat(out[b]['pred_scores'], 0)
        out[b]['pred_labels'] = torch.cat(out[b]['pred_labels'], 0) + 1
    return out


def recall_record(box_preds, gt_boxes, thresh_list):
    """generate_recall_record (detector3d_template.py:367-409) for one sample -> {'gt': n, 'rcnn_<t>': count}."""
    k = gt_boxes.shape[0] - 1
    while k >= 0 and float(gt_boxes[k].sum()) == 0:
        k -= 1
    gt = gt_boxes[:k + 1]
    rec = {'gt': int(gt.shape[0])}
    for t in thresh_list:
        rec['rcnn_%s' % t] = 0
    if gt.shape[0] and box_preds.shape[0]:
        iou = boxes_iou3d(box_preds[:, :7], gt[:, :7])
        for t in thresh_list:
            rec['rcnn_%s' % t] = int((iou.max(dim=0)[0] > t).sum())
    return rec
