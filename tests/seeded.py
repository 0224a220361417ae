"""Deterministic, name-keyed weight fill shared by the golden generator and the tests."""
import numpy as np
import torch


def seeded_fill_(state_dict, seed=0):
    """Deterministic, name-keyed fill of a state_dict (in place).

    Both the reference module (here) and the oracle / product modules (in tests) are filled with
    this function, so fixtures never need to carry weights.  Values depend only on
    (seed, key, shape).  BN running_var stays positive; integer buffers are left untouched.
    """
    import zlib
    for k in sorted(state_dict.keys()):
        v = state_dict[k]
        if not torch.is_floating_point(v):
            continue
        g = np.random.default_rng([seed, zlib.crc32(k.encode())])
        shape = tuple(v.shape)
        if k.endswith("running_var"):
            a = g.uniform(0.5, 1.5, size=shape)
        elif k.endswith("running_mean"):
            a = g.normal(0.0, 0.1, size=shape)
        elif k.endswith("norm.weight") or ".bn" in k and k.endswith("weight") or (
                v.dim() == 1 and k.endswith("weight")):
            a = g.uniform(0.5, 1.5, size=shape)
        elif v.dim() == 1 or k.endswith("bias") or "grn." in k:
            a = g.normal(0.0, 0.1, size=shape)
        else:
            fan_in = int(np.prod(shape[1:])) if v.dim() > 1 else shape[0]
            a = g.normal(0.0, 1.0, size=shape) * (1.0 / np.sqrt(max(fan_in, 1)))
        v.copy_(torch.from_numpy(np.asarray(a, dtype=np.float32)).reshape(shape))
    return state_dict
