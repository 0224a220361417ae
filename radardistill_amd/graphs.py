"""HIP-graph capture of the static-shape dense section of the training step.

The sparse stages (VFE, SparseEnc) have data-dependent row counts and stay eager.  Everything after `x_conv4.dense()` has
fixed shapes for a fixed (batch, grid): the teacher DenseEnc (+ head) and the student CMA + DenseEnc + CenterHead + target
assignment + AFD/PFD/detection losses and their whole backward.  That is ~80 % of the ~3300 kernel launches of a step; the
host cannot enqueue them as fast as the GPU runs them.  Captured once (after warm-up) and replayed per step, the host cost
of the section drops to two graph launches.  All librdamd entry points only enqueue on the given stream and take caller
memory, so they are capturable; caches that would hide weight re-layout kernels from the capture are bypassed while
capturing (autograd.kernel_weight), and zero-initialised scratch is allocated inside the capture (graph pool + fill node)
instead of the eager per-step arena.
"""
import torch
import torch.nn as nn

from . import autograd as A


class _TeacherDense(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.b2d, self.head, self.skip_head = model.backbone_2d, model.dense_head, model.skip_unused_teacher_head

    @torch.no_grad()
    def forward(self, x4, x5):
        bd = self.b2d({'multi_scale_2d_features': {'x_conv4': x4, 'x_conv5': x5}})
        outs = [bd['spatial_features_2d_8x'], bd['spatial_features_2d']]
        if not self.skip_head and self.head is not None:
            bd['batch_size'] = x4.shape[0]
            bd = self.head(bd)                    # lidar_pred_dicts: computed like the reference, consumed by nothing
            outs.append(bd['lidar_pred_dicts'][0]['hm'])
        return tuple(outs)


class _StudentDense(nn.Module):
    """CMA + DenseEnc + CenterHead + targets + distillation/detection losses -> [loss, tb values...] vector."""

    def __init__(self, model):
        super().__init__()
        self.r2d, self.rhead = model.radar_backbone_2d, model.radar_dense_head
        self.tb_names = None

    def forward(self, s4, s5, t4, l2d, l2d8, gt):
        bd = {'radar_multi_scale_2d_features': {'x_conv4': s4, 'x_conv5': s5}, 'multi_scale_2d_features': {'x_conv4': t4},
              'spatial_features_2d': l2d, 'spatial_features_2d_8x': l2d8, 'gt_boxes': gt, 'batch_size': s4.shape[0]}
        bd = self.rhead(self.r2d(bd))
        A.end_forward()                 # num_batches_tracked of this section's BatchNorms: part of the captured graph
        loss_feature, tb = self.r2d.get_loss(bd)
        loss_rpn, tb2 = self.rhead.get_loss()
        tb.update(tb2)
        loss = (loss_feature + loss_rpn).mean()
        self.tb_names = list(tb.keys())
        vals = torch.stack([v.detach().reshape(()).float() for v in tb.values()])
        return torch.cat([loss.reshape(1), vals])


class _GraphedInference:
    """Forward-only capture with static input/output buffers."""

    def __init__(self, module):
        self.module, self.graph, self.static_in, self.static_out, self.key = module, None, None, None, None

    def __call__(self, *args):
        key = tuple((tuple(a.shape), a.dtype) for a in args)
        if self.graph is None or key != self.key:
            self._capture(args, key)
        for s, a in zip(self.static_in, args):
            s.copy_(a)
        self.graph.replay()
        return self.static_out

    def _capture(self, args, key):
        self.static_in = [a.detach().clone() for a in args]
        with A.capture_scope():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.module(*self.static_in)
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_out = self.module(*self.static_in)
        self.key = key


class DenseSectionGraphs:
    def __init__(self, model):
        self.model = model
        self.teacher = _GraphedInference(_TeacherDense(model))
        self.student_module = _StudentDense(model)
        self.student = None
        self.key = None

    def _buffers(self):
        return [b for m in (self.student_module.r2d, self.student_module.rhead) for b in m.buffers()]

    def run(self, s4, s5, t4, t5, gt):
        touts = self.teacher(t4, t5)
        l2d8, l2d = touts[0], touts[1]
        args = (s4, s5, t4.detach(), l2d, l2d8, gt)
        key = tuple((tuple(a.shape), a.dtype, a.requires_grad) for a in args)
        if self.student is None or key != self.key:
            # make_graphed_callables runs 3 warm-up fwd+bwd and one capture pass: keep BatchNorm running statistics as they were
            saved = [b.detach().clone() for b in self._buffers()]
            self.student_module.train()
            sample = tuple(a.detach().clone().requires_grad_(a.requires_grad) for a in args)
            with A.capture_scope():     # warm-up + forward capture + backward capture (the latter on the autograd thread)
                self.student = torch.cuda.make_graphed_callables(self.student_module, sample, allow_unused_input=True)
            for b, s in zip(self._buffers(), saved):
                b.copy_(s)
            self.key = key
        out = self.student(*args)
        names = self.student_module.tb_names
        vals = out.detach()
        tb = {n: vals[1 + i] for i, n in enumerate(names)}
        return out[0], tb, (l2d8, l2d)
