"""The small training problem behind fixture g8_optim: a module tree, per-step gradients and schedule constants.

Shared by tests/golden/make_golden.py (which drives the REFERENCE's build_optimizer / OptimWrapper / OneCycle /
clip_grad_norm_ loop over it, tools/train_utils/train_utils.py:44-64) and by the tests (which drive the oracle and the fused
HIP optimizer over the very same tensors).  Own code: the tree only has to exercise what the reference optimizer
distinguishes --
  * BatchNorm leaves vs every other leaf (split_bn_bias, fastai_optim.py:16-28: two parameter groups),
  * a frozen parameter (trainable_params filters it out, fastai_optim.py:93-96),
  * a leaf module that holds bare Parameters (the GRN / DCN pattern),
  * parameters whose gradient is None on some steps (torch.optim.Adam skips them: no moment update, no step count;
    the decoupled decay of OptimWrapper.step still applies, fastai_optim.py:135-152),
  * steps whose gradient norm is below and above GRAD_NORM_CLIP,
  * a tensor larger than one 4096-element optimizer chunk of optim.hip.
"""
import numpy as np
import torch
import torch.nn as nn

TOTAL_ITERS_EACH_EPOCH, TOTAL_EPOCHS, N_STEPS = 4, 3, 8          # OneCycle over 12 steps, pct_start 0.4: both phases are visited
OPTIM_CFG = dict(OPTIMIZER='adam_onecycle', LR=0.001, WEIGHT_DECAY=0.01, MOMENTUM=0.9, MOMS=[0.95, 0.85], PCT_START=0.4,
                 DIV_FACTOR=10, DECAY_STEP_LIST=[35, 45], LR_DECAY=0.1, LR_CLIP=0.0000001, LR_WARMUP=False, WARMUP_EPOCH=1,
                 GRAD_NORM_CLIP=10)
GRAD_SIGMA = [0.05, 0.02, 0.5, 0.05, 1.5, 0.03, 0.2, 0.05]      # ~84 * sigma = total norm: steps 2, 4, 6 are clipped


class Scale(nn.Module):
    """Leaf module with bare Parameters (like Basicblock_convn.GRN)."""

    def __init__(self, n):
        super().__init__()
        self.gamma = nn.Parameter(torch.zeros(1, n))
        self.beta = nn.Parameter(torch.zeros(1, n))


class OptimCaseNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 8, 3, bias=True), nn.BatchNorm2d(8), nn.ReLU())
        self.body = nn.Sequential(nn.Linear(80, 64, bias=False), nn.BatchNorm1d(64, eps=1e-3, momentum=0.01), nn.ReLU(),
                                  nn.Sequential(nn.Linear(64, 16), nn.LayerNorm(16), Scale(16)))
        self.frozen = nn.Linear(16, 16)
        self.unused = nn.Linear(16, 4)          # never receives a gradient
        self.late = nn.Linear(16, 4)            # no gradient on step 0
        self.early = nn.Linear(16, 4)           # no gradient from step 3 on
        for p in self.frozen.parameters():
            p.requires_grad = False


def has_grad(name, step):
    if name.startswith("unused."):
        return False
    if name.startswith("late."):
        return step >= 1
    if name.startswith("early."):
        return step < 3
    return True


def grad_for(name, shape, step):
    import zlib
    g = np.random.default_rng([77, step, zlib.crc32(name.encode())])
    return torch.from_numpy(g.normal(0.0, GRAD_SIGMA[step], size=tuple(shape)).astype(np.float32))


def assign_grads(model, step):
    """p.grad for this step (None where has_grad says so); frozen parameters get none."""
    for name, p in model.named_parameters():
        if not p.requires_grad or not has_grad(name, step):
            p.grad = None
        else:
            p.grad = grad_for(name, p.shape, step).to(p.device)


def trainable(model):
    return [(n, p) for n, p in model.named_parameters() if p.requires_grad]
