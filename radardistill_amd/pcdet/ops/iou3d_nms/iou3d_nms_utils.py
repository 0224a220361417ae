"""pcdet/ops/iou3d_nms/iou3d_nms_utils.py on the HIP kernels: aligned 3-D IoU for the IoU-head loss (:83-117), rotated-BEV NMS
(:119-137) and pairwise 3-D IoU for the recall records (:55-81)."""
import torch

from radardistill_amd import kernels as K


def boxes_aligned_iou3d_gpu(boxes_a, boxes_b):
    """boxes (N,7) [x,y,z,dx,dy,dz,heading] -> (N,1)."""
    assert boxes_a.shape[0] == boxes_b.shape[0]
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    a_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(-1, 1)
    b_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(-1, 1)
    overlaps_bev = K.boxes_aligned_overlap_bev(boxes_a.detach().float().contiguous(), boxes_b.detach().float().contiguous())
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(-1, 1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """boxes_a (N,7), boxes_b (M,7) -> (N,M) 3-D IoU (iou3d_nms_utils.py:55-81)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    a_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1)
    b_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    overlaps_bev = K.boxes_overlap_bev(boxes_a.detach().float().contiguous(), boxes_b.detach().float().contiguous())
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """boxes (N,7), scores (N) -> (indices of the kept boxes in descending score order, None) (iou3d_nms_utils.py:119-137).
    The suppression matrix AND the greedy pass run on the device (rd_nms_bev); the only host read is the kept count."""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep, num = K.nms_bev(boxes[order].detach().float().contiguous(), thresh)
    return order[keep[:int(num.item())]].contiguous(), None
