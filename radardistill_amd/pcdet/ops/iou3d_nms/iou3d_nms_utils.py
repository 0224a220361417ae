"""Aligned 3-D IoU for the IoU-head loss (pcdet/ops/iou3d_nms/iou3d_nms_utils.py:83-117) on the HIP overlap kernel."""
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
