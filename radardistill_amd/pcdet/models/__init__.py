"""build_network / load_data_to_gpu / model_fn_decorator: the reference's training entry points
(pcdet/models/__init__.py:16-54)."""
import os
from collections import namedtuple

import numpy as np
import torch

from .detectors import build_detector


def build_network(model_cfg, num_class, dataset):
    model = build_detector(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
    if os.environ.get("RD_FAST_ATTRS", "1") != "0":
        # submodules and parameters mirrored into the instance dictionaries: `self.conv1` / `conv.weight` become plain attribute reads
        # instead of nn.Module.__getattr__ calls (~1200 per training step, -0.2 ms; autograd.fast_module_attrs says when it is safe)
        from radardistill_amd import autograd as A
        A.fast_module_attrs(model)
    return model


_COPY_STREAM = {}


def load_data_to_gpu(batch_dict):
    """numpy -> float32 CUDA tensors (reference: pageable `.cuda()` per key).  Host arrays go through pinned staging and
    non-blocking copies on a copy stream; `gt_boxes` additionally keeps a host copy (`gt_boxes_host`) so the CPU-side target
    assignment does not have to read it back.  batch_dict['_inputs_ready'] is an event recorded after the last copy: the
    detector's geometry prelude (detectors/pillarnet.py) waits for it instead of for the whole main stream."""
    cuda = torch.cuda.is_available()
    uploaded = []
    if cuda:
        dev = torch.cuda.current_device()
        cs = _COPY_STREAM.get(dev)
        if cs is None:
            cs = _COPY_STREAM[dev] = torch.cuda.Stream(dev)
    for key, val in list(batch_dict.items()):
        if not isinstance(val, np.ndarray):
            continue
        if key in ['frame_id', 'metadata', 'calib', 'image_paths', 'ori_shape', 'img_process_infos', 'gt_boxes_host']:
            continue
        if key == 'gt_boxes':
            batch_dict['gt_boxes_host'] = val
        if key in ['image_shape']:
            t = torch.from_numpy(val).int()
        else:
            t = torch.from_numpy(np.ascontiguousarray(val, dtype=np.float32))
        if cuda:
            with torch.cuda.stream(cs):
                t = t.pin_memory().cuda(non_blocking=True)
            uploaded.append(t)
        batch_dict[key] = t
    if uploaded:
        ev = torch.cuda.Event()
        ev.record(cs)
        main = torch.cuda.current_stream()
        main.wait_event(ev)
        for t in uploaded:
            t.record_stream(main)
        batch_dict['_inputs_ready'] = ev


def model_fn_decorator():
    ModelReturn = namedtuple('ModelReturn', ['loss', 'tb_dict', 'disp_dict'])

    def model_func(model, batch_dict):
        load_data_to_gpu(batch_dict)
        ret_dict, tb_dict, disp_dict = model(batch_dict)
        loss = ret_dict['loss'].mean()
        if hasattr(model, 'update_global_step'):
            model.update_global_step()
        else:
            model.module.update_global_step()
        return ModelReturn(loss, tb_dict, disp_dict)

    return model_func
