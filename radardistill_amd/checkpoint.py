"""Reading checkpoints into a detector and the recall bookkeeping of the eval loop -- the logic behind Detector3DTemplate's
`_load_state_dict`, `load_params_from_file`, `load_params_with_optimizer` and `generate_recall_record`
(pcdet/models/detectors/detector3d_template.py:367-496).  Contract kept: checkpoint dictionary keys (`model_state`, `optimizer_state`,
`epoch`, `it`, `version`), the `<name>_optim.<ext>` side file, non-strict loading by name AND shape, the sparse-kernel layout fix-up for
files written by spconv 1.x, the `recall_dict` keys (`gt`, `roi_<t>`, `rcnn_<t>`).  Files are read with `weights_only=True`."""
import os

import torch


def read_checkpoint(path, to_cpu=False):
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    return torch.load(path, map_location=torch.device('cpu') if to_cpu else None, weights_only=True)


def optimizer_side_file(path):
    """`foo.pth` -> `foo_optim.pth`: where the reference's loop may have parked the optimizer state (detector3d_template.py:484-490)."""
    stem, ext = os.path.splitext(path)
    if len(ext) != 4:
        raise ValueError(f"checkpoint file name {path!r} needs a three-letter extension")
    return f"{stem}_optim{ext}"


def _legacy_sparse_layouts(t):
    """Candidate re-layouts of a sparse-conv kernel stored by another spconv generation: 1.x kept (k1, k2, c_in, c_out) where 2.x and this
    build keep (c_out, k1, k2, c_in); some 1.x builds had the last two axes the other way round."""
    yield t.transpose(-1, -2)
    if t.dim() == 4:
        yield t.permute(3, 0, 1, 2)


def fit_state_to_model(model, disk_state, sparse_keys):
    """{name: tensor} for every entry of `disk_state` that the model has under the same name and -- possibly after a legacy sparse-kernel
    re-layout -- the same shape.  Everything else is left out (a student initialised from LiDAR weights silently skips the radar VFE's
    15-column Linear this way, ckpt.py)."""
    own = model.state_dict()
    fitted = {}
    for name, value in disk_state.items():
        want = own.get(name)
        if want is None:
            continue
        if value.shape != want.shape and name in sparse_keys:
            value = next((c.contiguous() for c in _legacy_sparse_layouts(value) if c.shape == want.shape), value)
        if value.shape == want.shape:
            fitted[name] = value
    return own, fitted


def count_recalled(iou_pred_gt, thresholds):
    """iou (n_pred, n_gt) -> how many ground-truth boxes have a prediction above each threshold, all thresholds in ONE device
    comparison and one host read (the reference reads one scalar per threshold)."""
    if iou_pred_gt.shape[0] == 0:
        return [0] * len(thresholds)
    best = iou_pred_gt.max(dim=0)[0]
    th = torch.as_tensor(list(thresholds), dtype=best.dtype, device=best.device)
    return (best[None, :] > th[:, None]).sum(dim=1).tolist()


def strip_padding(gt_boxes):
    """Ground-truth rows up to the last one that is not all zero (collate_batch zero-pads to the longest sample)."""
    used = (gt_boxes.sum(dim=1) != 0).nonzero()
    return gt_boxes[:int(used.max().item()) + 1] if used.numel() else gt_boxes[:0]
