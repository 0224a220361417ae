"""Oracle: optimizer step of the reference training loop.  Test infrastructure.

Follows
  tools/train_utils/train_utils.py:55-64                      (zero_grad, backward, clip_grad_norm_(10), step)
  tools/train_utils/optimization/__init__.py:19-33            (Adam betas=(0.9,0.99), OptimWrapper true_wd, bn_wd)
  tools/train_utils/optimization/fastai_optim.py:135-152      (p *= 1 - wd*lr for every trainable param, then Adam with wd=0)
  tools/train_utils/optimization/learning_schedules_fastai.py:44-77 (OneCycle: cosine lr and momentum phases)
"""
import math

import numpy as np
import torch


def annealing_cos(start, end, pct):
    return end + (start - end) / 2 * (np.cos(np.pi * pct) + 1)


def one_cycle(step, total_step, lr_max=1e-3, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4):
    """(lr, beta1) at iteration `step` (learning_schedules_fastai.py:44-77)."""
    low = lr_max / div_factor
    p1 = int(total_step * pct_start)
    lr_ph = [(0, p1, (low, lr_max)), (p1, total_step, (lr_max, low / 1e4))]
    mom_ph = [(0, p1, (moms[0], moms[1])), (p1, total_step, (moms[1], moms[0]))]
    lr, mom = low, moms[0]
    for s, e, (a, b) in lr_ph:
        if step >= s:
            lr = annealing_cos(a, b, (step - s) / (e - s))
    for s, e, (a, b) in mom_ph:
        if step >= s:
            mom = annealing_cos(a, b, (step - s) / (e - s))
    return float(lr), float(mom)


def clip_grad_norm(grads, max_norm=10.0):
    """torch.nn.utils.clip_grad_norm_ (train_utils.py:62): total 2-norm over the gradients that exist (None entries are skipped);
    every gradient is scaled by max_norm/(norm+1e-6) clamped to 1."""
    present = [g for g in grads if g is not None]
    total = torch.sqrt(sum((g.detach().double() ** 2).sum() for g in present)).float() if present else torch.zeros(())
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, [None if g is None else g * coef for g in grads]


def adam_true_wd_step(params, grads, exp_avg, exp_avg_sq, steps, lr, beta1, beta2=0.99, wd=0.01, eps=1e-8):
    """One OptimWrapper.step() (fastai_optim.py:135-152): decoupled decay `p *= 1 - wd*lr` on EVERY trainable parameter, then
    torch.optim.Adam (no amsgrad, weight_decay 0), which skips a parameter whose gradient is None -- no moment update and no step
    count for it.  In place.  `steps[i]` is parameter i's own count of Adam updates so far (an int `steps` = the same count for
    all, incremented here into nothing: the caller passes the 1-based count AFTER this update, as before)."""
    per_param = not isinstance(steps, int)
    for i, (p, g, m, v) in enumerate(zip(params, grads, exp_avg, exp_avg_sq)):
        p.mul_(1 - wd * lr)
        if g is None:
            continue
        if per_param:
            steps[i] += 1
            step = steps[i]
        else:
            step = steps
        bc1 = 1 - beta1 ** step
        bc2 = 1 - beta2 ** step
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)
