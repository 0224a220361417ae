"""`spconv` namespace + helpers with the reference's names (pcdet/utils/spconv_utils.py:1-38), backed by
radardistill_amd.sparse (the real spconv is an absent third-party CUDA library)."""
from typing import Set

import torch.nn as nn

from radardistill_amd import sparse as spconv


def find_all_spconv_keys(model: nn.Module, prefix="") -> Set[str]:
    found: Set[str] = set()
    for name, child in model.named_children():
        new_prefix = f"{prefix}.{name}" if prefix != "" else name
        if isinstance(child, spconv.conv.SparseConvolution):
            found.add(f"{new_prefix}.weight")
        found.update(find_all_spconv_keys(child, prefix=new_prefix))
    return found


def replace_feature(out, new_features):
    return out.replace_feature(new_features)
