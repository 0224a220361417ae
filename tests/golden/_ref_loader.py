"""Load *leaf files* of the read-only reference tree on CPU, for golden-vector generation only.

This module is used ONLY by tests/golden/make_golden.py, in the build container where
/root/reference exists.  Nothing on the GPU box imports it (the reference does not travel).

Why a loader is needed: `import pcdet` fails wholesale (missing generated
version.py, spconv, compiled extensions, cv2, numba, easydict ...).  Individual leaf files are
importable when
  * the package __init__ files are bypassed by pre-seeding `sys.modules` with empty namespace
    packages that only carry a `__path__`,
  * absent third-party modules are replaced by inert placeholders (they are never *called* by the
    code paths we run, except `torch_scatter`, see below),
  * `Tensor.cuda()` is made the identity (constructors call `.cuda()` on small constant tensors).

`torch_scatter` is not installed; the two functions the VFE calls are supplied from plain torch
(`index_add_`/`scatter_reduce_`).  Goldens of the VFE therefore pin everything in that file except
the third-party scatter arithmetic itself (segment mean / max; "parity unpinned" for that library).
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF_ROOT = os.environ.get("RADARDISTILL_REFERENCE", "/root/reference")


class AttrDict(dict):
    """Minimal EasyDict stand-in (easydict is not installed here)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = self._wrap(v)


def _ns(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    m.__package__ = name
    sys.modules[name] = m
    return m


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _scatter_mean(src, index, dim=0):
    n = int(index.max()) + 1 if index.numel() else 0
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    out.index_add_(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype)
    cnt.index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
    return out / cnt.clamp(min=1).unsqueeze(-1)


def _scatter_max(src, index, dim=0):
    n = int(index.max()) + 1 if index.numel() else 0
    out = torch.full((n,) + tuple(src.shape[1:]), float("-inf"), dtype=src.dtype)
    idx = index.unsqueeze(-1).expand_as(src)
    out = out.scatter_reduce(0, idx, src, reduce="amax", include_self=True)
    return out, None


def install():
    """Prepare sys.modules so reference leaf files can be imported by dotted name."""
    if "pcdet" in sys.modules and getattr(sys.modules["pcdet"], "_rd_shell", False):
        return
    p = os.path.join(REF_ROOT, "pcdet")
    shells = {
        "pcdet": p,
        "pcdet.models": f"{p}/models",
        "pcdet.models.backbones_2d": f"{p}/models/backbones_2d",
        "pcdet.models.backbones_3d": f"{p}/models/backbones_3d",
        "pcdet.models.backbones_3d.vfe": f"{p}/models/backbones_3d/vfe",
        "pcdet.models.dense_heads": f"{p}/models/dense_heads",
        "pcdet.models.model_utils": f"{p}/models/model_utils",
        "pcdet.utils": f"{p}/utils",
        "pcdet.ops": f"{p}/ops",
        "pcdet.ops.basicblock": f"{p}/ops/basicblock",
        "pcdet.ops.basicblock.modules": f"{p}/ops/basicblock/modules",
        "pcdet.ops.iou3d_nms": f"{p}/ops/iou3d_nms",
    }
    for name, path in shells.items():
        _ns(name, path)
    sys.modules["pcdet"]._rd_shell = True

    # absent third-party deps: inert placeholders
    _stub("cv2")
    _stub("numba", jit=lambda *a, **k: (lambda f: f))
    _stub("SharedArray")
    _stub("openpyxl")
    _stub("torch_scatter", scatter_mean=_scatter_mean, scatter_max=_scatter_max)
    # compiled extensions / heavy utils that the leaf files import but our paths never call
    _stub("pcdet.ops.iou3d_nms.iou3d_nms_cuda")
    _stub("pcdet.utils.common_utils", check_numpy_to_torch=lambda x: (x, False))
    _stub("pcdet.utils.box_utils", center_to_corner_box2d=None)
    _stub("pcdet.models.model_utils.model_nms_utils")
    torch.Tensor.cuda = lambda self, *a, **k: self  # constructors call .cuda() on constants


def load(dotted):
    install()
    return importlib.import_module(dotted)
