"""Detection losses of the CenterHead with the reference's names and values
(pcdet/utils/loss_utils.py:266-301 neg_loss_cornernet, :347-419 _reg_loss / RegLossCenterNet, :379-394 gather helpers,
:651-673 IouLoss, :677-701 IouRegLoss).

Re-designed to stay on the device: the reference branches on the host (`if num_pos == 0`, `if mask.sum() == 0`) and
boolean-indexes with data-dependent sizes -- each a blocking device->host sync, ~3 per head.  Here the same values are
produced with masked arithmetic over the fixed (B, 500) object slots (masked-out slots contribute exact zeros), so a
training step has no sync in its loss.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..models.model_utils.centernet_utils import bbox3d_overlaps_diou
from ..ops.iou3d_nms import iou3d_nms_utils


def neg_loss_cornernet(pred, gt, mask=None):
    pos_inds = gt.eq(1).float()
    neg_inds = gt.lt(1).float()
    neg_weights = torch.pow(1 - gt, 4)
    pos_loss = torch.log(pred) * torch.pow(1 - pred, 2) * pos_inds
    neg_loss = torch.log(1 - pred) * torch.pow(pred, 2) * neg_weights * neg_inds
    if mask is not None:
        mask = mask[:, None, :, :].float()
        pos_loss = pos_loss * mask
        neg_loss = neg_loss * mask
        num_pos = (pos_inds.float() * mask).sum()
    else:
        num_pos = pos_inds.float().sum()
    pos_loss = pos_loss.sum()
    neg_loss = neg_loss.sum()
    # num_pos == 0  ->  -neg_loss  (pos_loss is an exact 0 then), else -(pos + neg) / num_pos: no host branch
    return -(pos_loss + neg_loss) / torch.clamp_min(num_pos, 1.0)


class FocalLossCenterNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.neg_loss = neg_loss_cornernet

    def forward(self, out, target, mask=None):
        return self.neg_loss(out, target, mask=mask)


def _reg_loss(regr, gt_regr, mask):
    num = mask.float().sum()
    mask = mask.unsqueeze(2).expand_as(gt_regr).float()
    isnotnan = (~torch.isnan(gt_regr)).float()
    mask = mask * isnotnan
    regr = regr * mask
    gt_regr = gt_regr * mask
    loss = torch.abs(regr - gt_regr)
    loss = loss.transpose(2, 0)
    loss = torch.sum(loss, dim=2)
    loss = torch.sum(loss, dim=1)
    return loss / torch.clamp_min(num, min=1.0)


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


class RegLossCenterNet(nn.Module):
    def forward(self, output, mask, ind=None, target=None):
        pred = output if ind is None else _transpose_and_gather_feat(output, ind)
        return _reg_loss(pred, target, mask)


_SAFE_BOX = None


def _masked_boxes(boxes, mask_b):
    """Replace masked-out (all-zero gt / arbitrary pred) slots by a harmless unit box so no NaN/inf reaches the masked sum."""
    from radardistill_amd.autograd import const_tensor
    safe = const_tensor("safe_box", [0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 0.0], boxes.device, boxes.dtype)
    return torch.where(mask_b.unsqueeze(-1), boxes, safe.expand_as(boxes))


class IouLoss(nn.Module):
    """L1 between the IoU head and 2*IoU3D-1 of (detached) decoded boxes vs gt, over positive slots."""

    def forward(self, iou_pred, mask, ind, box_pred, box_gt):
        mb = mask.bool()
        n = mask.sum()
        pred = _transpose_and_gather_feat(iou_pred, ind)                       # (B, K, 1)
        pred_box = _masked_boxes(_transpose_and_gather_feat(box_pred, ind), mb)   # (B, K, 7)
        gt = _masked_boxes(box_gt[..., :7], mb)
        B, Kk = mask.shape
        target = iou3d_nms_utils.boxes_aligned_iou3d_gpu(pred_box.reshape(-1, 7), gt.reshape(-1, 7)).view(B, Kk, 1)
        target = 2 * target - 1
        loss = (torch.abs(pred - target) * mb.unsqueeze(-1).float()).sum()
        # reference: zeros(1) when there is no positive, else sum / (n + 1e-4): the masked sum is already an exact 0 then
        return (loss / (n + 1e-4)).view(1)


class IouRegLoss(nn.Module):
    def __init__(self, type="IoU"):
        super().__init__()
        if type == "DIoU":
            self.bbox3d_iou_func = bbox3d_overlaps_diou
        else:
            raise NotImplementedError

    def forward(self, box_pred, mask, ind, box_gt):
        mb = mask.bool()
        n = mask.sum()
        pred_box = _masked_boxes(_transpose_and_gather_feat(box_pred, ind), mb)
        gt = _masked_boxes(box_gt[..., :7], mb)
        B, Kk = mask.shape
        iou = self.bbox3d_iou_func(pred_box.reshape(-1, 7), gt.reshape(-1, 7)).view(B, Kk)
        loss = ((1. - iou) * mb.float()).sum() / (n + 1e-4)
        return loss.view(1)
