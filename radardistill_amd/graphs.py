"""HIP-graph replay of the static-shape dense section of the distillation step (MODEL.DENSE_GRAPH: True / RD_DENSE_GRAPH=1).

Why: at B = 8 the Python host needs ~20 ms to enqueue the ~1200 launches of a step while the GPU needs ~21 ms to run them; the
host is co-critical.  Everything after `x_conv4.dense()` has shapes fixed by (batch, grid): both branches' conv5, the teacher's
DenseEnc (+ head), the student's CMA + DenseEnc + CenterHead, target assignment, AFD / PFD / detection losses and the whole
backward of the student half -- ~80 % of the launches.  That section is captured ONCE into one HIP graph (forward + loss + backward,
through torch's stream capture: every librdamd entry point only enqueues on the stream it is given and takes caller memory, so
its launches are capturable) and replayed per step with one call; VFE / SparseEnc (data-dependent row counts) stay eager and
receive d loss / d x_conv4 from the replay.

Ownership rule (round 1's prototype broke it): a captured launch may only touch memory the capture owns or
memory that is persistent for the life of the graph.
  * inputs are copied into static buffers (teacher x_conv4, student x_conv4, gt_boxes padded to a fixed row count);
  * zero-filled scratch and gradient accumulators come from torch.zeros inside the capture (autograd.CAPTURING), never from the
    eager per-step arenas; version-keyed host caches of re-laid-out weights are bypassed; the operand cache's persistent buffers
    are refreshed eagerly by begin_step() before every replay, as in the eager path;
  * nothing is uploaded from the host inside the capture (constants are created by two eager warm-up passes first);
  * the side streams (teacher branch, weight gradients) fork from and join the capturing stream inside the capture.
Parameter gradients come out of the replay in static tensors; the optimizer's descriptor table therefore never changes.
Not supported in this mode (raises): gradient accumulation across several backward passes, double backward.
"""
import os

import torch

from . import autograd as A
from . import kernels as K


def enabled(model_cfg):
    env = os.environ.get("RD_DENSE_GRAPH")
    if env is not None:
        return env == "1"
    return bool(model_cfg.get("DENSE_GRAPH", False))


class _Replay(torch.autograd.Function):
    """Eager-side node: forward = copy inputs + one graph launch (which already ran the section's backward); backward hands the
    stored d loss / d x_conv4 to the sparse encoder and publishes the parameter gradients."""

    @staticmethod
    def forward(ctx, s4, section):
        ctx.section = section
        return section._replay(s4)

    @staticmethod
    def backward(ctx, g_loss, _g_tb):
        return ctx.section._publish(g_loss), None


class DenseSection:
    def __init__(self, model):
        self.model = model
        self.graph = None
        self.key = None

    # ---- the section as plain module calls (used for warm-up and capture)
    def _run(self, t4, s4, gt):
        m = self.model
        B = s4.shape[0]
        dev = s4.device
        main = torch.cuda.current_stream(dev)
        bd = {'batch_size': B, 'gt_boxes': gt}
        side = m._teacher_stream if getattr(m, '_teacher_stream', None) is not None else None
        use_side = side is not None and m.model_cfg.get('TEACHER_STREAM', True) and os.environ.get('RD_TEACHER_STREAM', '1') != '0'
        frozen = [m.backbone_3d, m.backbone_2d, m.dense_head]
        for mod in frozen:
            if mod is not None and mod.training:
                mod.eval()

        def teacher():
            with torch.no_grad():
                t5 = m.backbone_3d.conv5(t4)
                bd['multi_scale_2d_features'] = {'x_conv4': t4, 'x_conv5': t5}
                m.backbone_2d(bd)
                if m.dense_head is not None and not m.skip_unused_teacher_head:
                    m.dense_head(bd)

        if use_side:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                teacher()
        else:
            teacher()
        s5 = m.radar_backbone_3d.conv5(s4)
        bd['radar_multi_scale_2d_features'] = {'x_conv4': s4, 'x_conv5': s5}
        m.radar_backbone_2d(bd)
        m.radar_dense_head(bd)
        if use_side:
            main.wait_stream(side)
        A.end_forward()
        loss, tb, _ = m.get_training_distll_loss(bd)
        loss = loss.mean()
        names = list(tb.keys())
        vals = torch.stack([torch.as_tensor(v).detach().reshape(()).float() for v in tb.values()])
        return loss, names, vals

    def _params(self):
        m = self.model
        mods = [m.radar_backbone_3d.conv5, m.radar_backbone_2d, m.radar_dense_head]
        return [p for mod in mods for p in mod.parameters() if p.requires_grad]

    def _buffers(self):
        m = self.model
        mods = [m.radar_backbone_3d.conv5, m.radar_backbone_2d, m.radar_dense_head]
        return [b for mod in mods for b in mod.buffers()]

    def _capture(self, t4, s4, gt):
        dev = s4.device
        self.st_t4 = torch.empty_like(t4)
        self.st_s4 = torch.empty_like(s4).requires_grad_(True)
        self.st_gt = torch.zeros_like(gt)
        self.st_t4.copy_(t4); self.st_gt.copy_(gt)
        with torch.no_grad():
            self.st_s4.copy_(s4)
        self.params = self._params()
        saved = [b.detach().clone() for b in self._buffers()]
        prev_wg = A.WGRAD_STREAM[0]
        A.CAPTURING[0] = True
        try:
            # two eager passes on a side stream first: every constant / operand-cache entry / function attribute the section creates on
            # first use exists before the capture, and the allocator's private pool sees the same request sequence it will capture
            cur = torch.cuda.current_stream(dev)
            warm = torch.cuda.Stream(dev)
            warm.wait_stream(cur)
            with torch.cuda.stream(warm):
                for _ in range(2):
                    loss, names, vals = self._run(self.st_t4, self.st_s4, self.st_gt)
                    torch.autograd.grad(loss, [self.st_s4] + self.params, allow_unused=True)
            cur.wait_stream(warm)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                loss, names, vals = self._run(self.st_t4, self.st_s4, self.st_gt)
                grads = torch.autograd.grad(loss, [self.st_s4] + self.params, allow_unused=True)
        finally:
            A.CAPTURING[0] = False
            A.WGRAD_STREAM[0] = prev_wg
        with torch.no_grad():                      # the warm-up passes moved the BatchNorm running statistics; the capture itself ran nothing
            for b, s in zip(self._buffers(), saved):
                b.copy_(s)
        self.out_loss, self.tb_names, self.out_tb = loss.detach(), names, vals
        self.g_s4 = grads[0]
        self.g_params = [(p, g) for p, g in zip(self.params, grads[1:]) if g is not None]
        self._flat_grads = [g for _, g in self.g_params]

    # ---- per step
    def run(self, t4, s4, gt):
        """t4: teacher x_conv4 (no grad); s4: student x_conv4 (in the autograd graph of the sparse encoder); gt: (B, M, 10) device
        tensor.  -> (loss, tb_dict) with loss connected to s4."""
        M = gt.shape[1]
        key = (tuple(t4.shape), tuple(t4.stride()), tuple(s4.shape), tuple(s4.stride()), gt.shape[0], gt.shape[2], K.get_conv_math(),
               K.get_deterministic(), A.WGRAD_STREAM[0], os.environ.get('RD_TEACHER_STREAM', '1'))
        if self.graph is None or key != self.key or M > self.st_gt.shape[1]:
            cap = max(M, 64)
            pad = torch.zeros((gt.shape[0], cap, gt.shape[2]), dtype=gt.dtype, device=gt.device)
            pad[:, :M] = gt
            self._capture(t4.detach(), s4.detach(), pad)
            self.key = key
        self.st_t4.copy_(t4.detach())
        if M == self.st_gt.shape[1]:
            self.st_gt.copy_(gt)
        else:
            self.st_gt.zero_()
            self.st_gt[:, :M].copy_(gt)          # all-zero rows are padding: class id 0 belongs to no head
        loss, tb = _Replay.apply(s4, self)
        return loss, {n: tb[i] for i, n in enumerate(self.tb_names)}

    def _replay(self, s4):
        with torch.no_grad():
            self.st_s4.copy_(s4)
        self.graph.replay()
        return self.out_loss.clone(), self.out_tb.clone()

    def _publish(self, g_loss):
        """Called from the eager backward: parameter gradients of the section (computed by the replay) become .grad."""
        scale = g_loss.reshape(())
        torch._foreach_mul_(self._flat_grads + [self.g_s4], scale)          # d(total) / d(section loss); 1 for a plain loss.backward()
        for p, g in self.g_params:
            if p.grad is None:
                p.grad = g
            elif p.grad.data_ptr() == g.data_ptr():
                raise RuntimeError("DENSE_GRAPH: parameter gradients live in the graph's static buffers; call zero_grad() between "
                                   "backward passes (gradient accumulation is not supported in this mode)")
            else:
                p.grad.add_(g)
        return self.g_s4
