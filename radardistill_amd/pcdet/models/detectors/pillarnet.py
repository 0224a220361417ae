"""PillarNet with the distillation branches (pcdet/models/detectors/pillarnet.py:12-96): freeze list by class name,
module chain over a shared batch_dict, three loss modes."""
import os

import torch

from .detector3d_template import Detector3DTemplate


class PillarNet(Detector3DTemplate):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()
        self.model_cfg = model_cfg
        if self.model_cfg.get('FREEZE_PIPELINE', None) is not None:
            self.no_grad_module = model_cfg['FREEZE_PIPELINE']
            for cur_module in self.module_list:
                if cur_module.__class__.__name__ in self.no_grad_module:
                    for param in cur_module.parameters():
                        param.requires_grad = False
        else:
            self.no_grad_module = []
        # The frozen teacher CenterHead's predictions (`lidar_pred_dicts`) are consumed by nothing in the distill loss
        # (pillarnet.py:65-73).  Kept on by default for fidelity; MODEL.SKIP_UNUSED_TEACHER_HEAD: True drops that dead work.
        self.skip_unused_teacher_head = bool(self.model_cfg.get('SKIP_UNUSED_TEACHER_HEAD', False))

    def _forward_dense_graph(self, batch_dict, dev):
        """Training forward with the static dense section replayed as one HIP graph (radardistill_amd/graphs.py): VFE and the four
        sparse stages of both branches run eagerly as in forward(), everything after `x_conv4.dense()` -- conv5, DenseEnc, CMA,
        heads, targets, losses and their backward -- is one graph launch."""
        from radardistill_amd import autograd as A
        from radardistill_amd import graphs as G
        A.begin_step(dev)
        if os.environ.get('RD_GEOM_STREAM', '1') != '0':
            self._geometry_prelude(batch_dict, dev)
        main = torch.cuda.current_stream(dev)
        fork = self.model_cfg.get('TEACHER_STREAM', True) and os.environ.get('RD_TEACHER_STREAM', '1') != '0'
        if fork and getattr(self, '_teacher_stream', None) is None:
            self._teacher_stream = torch.cuda.Stream(dev, priority=int(os.environ.get('RD_TEACHER_PRIO', '0')))
        side = self._teacher_stream if fork else None
        for m in (self.vfe, self.backbone_3d, self.backbone_2d, self.dense_head):
            if m is not None and m.training:
                m.eval()
        with torch.no_grad():
            batch_dict = self.vfe(batch_dict)
        batch_dict = self.radar_vfe(batch_dict)
        self.backbone_3d.prepare(batch_dict)
        self.radar_backbone_3d.prepare(batch_dict)
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side), torch.no_grad():
                t4 = self.backbone_3d.forward_sparse(batch_dict)
        else:
            with torch.no_grad():
                t4 = self.backbone_3d.forward_sparse(batch_dict)
        s4 = self.radar_backbone_3d.forward_sparse(batch_dict)
        if side is not None:
            main.wait_stream(side)
            t4.record_stream(main)
        A.end_forward()                       # the sparse stages' BatchNorm counters; the section's own are part of the graph
        if getattr(self, '_dense_section', None) is None:
            self._dense_section = G.DenseSection(self)
        loss, tb_dict = self._dense_section.run(t4, s4, batch_dict['gt_boxes'])
        return {'loss': loss}, tb_dict, {}

    def forward(self, batch_dict):
        from radardistill_amd import autograd as A
        dev = next(self.parameters()).device
        if dev.type == "cuda" and self.training and self.model_cfg.get('DISTILL', None) and '_teacher_done' not in batch_dict:
            from radardistill_amd import graphs as G
            if G.enabled(self.model_cfg):
                return self._forward_dense_graph(batch_dict, dev)
        if dev.type == "cuda":
            A.begin_step(dev)
        prepared = False
        teacher_done = batch_dict.pop('_teacher_done', None)        # set by prefetch_teacher(): frozen modules already ran for this batch
        if dev.type == "cuda" and os.environ.get('RD_GEOM_STREAM', '1') != '0' and teacher_done is None:
            self._geometry_prelude(batch_dict, dev)
        # Frozen teacher on its own HIP stream (training only): after the rulebook pyramids exist, the teacher's backbone / DenseEnc /
        # head are enqueued on a side stream while the student's modules go to the main stream, so the many kernels of either branch
        # that do not fill 256 CUs (sparse stages, BatchNorm passes, 64x64-tile convs) overlap.  The student never reads teacher
        # tensors before the losses; the streams join right before them.  MODEL.TEACHER_STREAM: False / RD_TEACHER_STREAM=0 disables.
        fork = (dev.type == "cuda" and self.training and bool(self.no_grad_module) and self.model_cfg.get('TEACHER_STREAM', True)
                and os.environ.get('RD_TEACHER_STREAM', '1') != '0')
        main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        side, forked = None, False
        for cur_module in self.module_list:
            cur_name = cur_module.__class__.__name__
            if not prepared and hasattr(cur_module, 'prepare'):
                # Build the active-site pyramids (rulebooks) of BOTH branches now, while the stream only holds the cheap VFE
                # kernels: their device->host count read-backs then never wait behind convolution work.
                for m in self.module_list:
                    if hasattr(m, 'prepare') and not (teacher_done is not None and m.__class__.__name__ in self.no_grad_module):
                        m.prepare(batch_dict)
                prepared = True
            if cur_name in self.no_grad_module:
                if teacher_done is not None:
                    continue
                if cur_module.training:           # (model.train() re-arms it; the recursive .eval() costs 1.6 ms/step if done blindly)
                    cur_module.eval()
                if self.skip_unused_teacher_head and cur_name == 'CenterHead' and self.training:
                    continue
                if fork and prepared:
                    if not forked:
                        if getattr(self, '_teacher_stream', None) is None:
                            self._teacher_stream = torch.cuda.Stream(dev, priority=int(os.environ.get('RD_TEACHER_PRIO', '0')))
                        side = self._teacher_stream
                        side.wait_stream(main)          # inputs, rulebooks (and last step's readers of recycled teacher memory) are done
                        forked = True
                    with torch.cuda.stream(side), torch.no_grad():
                        batch_dict = cur_module(batch_dict)
                else:
                    with torch.no_grad():          # frozen modules: fused inference kernels, no autograd graph
                        batch_dict = cur_module(batch_dict)
            else:
                batch_dict = cur_module(batch_dict)
        if forked:
            main.wait_stream(side)                  # the losses read teacher feature maps
        if teacher_done is not None:
            main.wait_event(teacher_done)
        if dev.type == "cuda":
            A.end_forward()
        if self.training:
            if self.model_cfg.get('DISTILL', None) is None:
                loss, tb_dict, disp_dict = self.get_training_loss()
            elif self.model_cfg.get('DISTILL', None):
                loss, tb_dict, disp_dict = self.get_training_distll_loss(batch_dict)
            else:
                loss, tb_dict, disp_dict = self.get_training_wo_distll_loss(batch_dict)
            return {'loss': loss}, tb_dict, disp_dict
        return self.post_processing(batch_dict)

    def prefetch_teacher(self, batch_dict):
        """Software pipelining across steps (optional; bench.py and train.py call it between backward and optimizer.step for the NEXT
        batch): the index work of both branches and every frozen (teacher) module run now, on the geometry / teacher streams, and
        their outputs wait in `batch_dict` for the forward() that later receives this same dict.  The teacher does not depend on the
        weights the optimizer is about to update, and its stream first waits for the main stream, so its ~5 ms of dense kernels land
        exactly where the main stream is launch-bound and the GPU used to idle: the tail of backward, the optimizer, and the sparse
        start of the next forward (kernel trace: 3-4 ms of idle gaps per step around the step boundary).  Work per step is unchanged
        -- one teacher forward, one student forward + backward -- only its placement moves.  No-op (the dict is returned untouched)
        when the teacher stream is off, in eval mode, or on the CPU.
        OFF by default (RD_TEACHER_PREFETCH=1 enables): measured neutral (308 vs 312 samples/s).  With the earlier four-read index
        prelude it lost 8 %: the read-backs then happen while the GPU is saturated by the backward pass -- each waited 0.3-0.5 ms for
        a free wave slot even on the high-priority stream -- and the host is the scarcer resource at B = 8."""
        from radardistill_amd.pcdet.models import load_data_to_gpu
        dev = next(self.parameters()).device
        if not (dev.type == "cuda" and self.training and bool(self.no_grad_module) and self.model_cfg.get('TEACHER_STREAM', True)
                and os.environ.get('RD_TEACHER_STREAM', '1') != '0' and os.environ.get('RD_GEOM_STREAM', '1') != '0'
                and os.environ.get('RD_TEACHER_PREFETCH', '0') == '1'):
            return batch_dict
        load_data_to_gpu(batch_dict)
        main = torch.cuda.current_stream(dev)
        self._geometry_prelude(batch_dict, dev)
        side = self._teacher_stream
        side.wait_stream(main)        # after everything enqueued so far (the backward pass): the teacher fills the idle boundary
        before = {k: v for k, v in batch_dict.items()}
        with torch.cuda.stream(side), torch.no_grad():
            for cur_module in self.module_list:
                cur_name = cur_module.__class__.__name__
                if cur_name not in self.no_grad_module:
                    continue
                if cur_module.training:
                    cur_module.eval()
                if self.skip_unused_teacher_head and cur_name == 'CenterHead':
                    continue
                if hasattr(cur_module, 'prepare'):
                    cur_module.prepare(batch_dict)
                batch_dict = cur_module(batch_dict)
            done = torch.cuda.Event()
            done.record(side)

        def _record(v):
            if torch.is_tensor(v):
                if v.is_cuda:
                    v.record_stream(main)
            elif isinstance(v, dict):
                for x in v.values():
                    _record(x)
            elif isinstance(v, (list, tuple)):
                for x in v:
                    _record(x)

        for k, v in batch_dict.items():              # teacher outputs live in the teacher stream's pool and are read on the main stream
            if k not in before or before[k] is not v:
                _record(v)
            elif torch.is_tensor(v) and v.is_cuda:
                v.record_stream(side)                # inputs (points, gt boxes) were read on the teacher stream
        batch_dict['_teacher_done'] = done
        return batch_dict

    def _geometry_prelude(self, batch_dict, dev):
        """All index work of the step -- voxelisation and the 4-level active-site pyramids of BOTH branches -- on its own
        high-priority HIP stream, before anything else is enqueued.  This work needs the host (array sizes = active-site counts read
        back from the device: ONE read covering both branches) but depends only on the input points, not on the previous
        step.  On the main stream every such read drained the whole queue -- the host could never enqueue ahead of the GPU, and
        the GPU starved through the launch-bound forward (measured: 34.3 ms/step against 25 ms of pure host enqueue time and
        ~27 ms of kernels).  Here the reads only wait for a few small kernels while the main stream is still busy with the previous
        step's backward / optimizer.
        Inputs must be complete when the prelude starts: batch_dict['_inputs_ready'] (a torch.cuda.Event recorded after the
        upload; load_data_to_gpu and bench.py provide it) is waited for on the geometry stream; without it the prelude waits for
        the main stream (correct for any caller, no overlap)."""
        from radardistill_amd import sparse as SP
        vfes = [m for m in self.module_list if hasattr(m, 'geometry_begin') and m.POINTS_KEY in batch_dict]
        if not vfes:
            return
        main = torch.cuda.current_stream(dev)
        gs = getattr(self, '_geom_stream', None)
        if gs is None:
            gs = self._geom_stream = torch.cuda.Stream(dev, priority=-1)
        ready = batch_dict.get('_inputs_ready', None)
        with torch.cuda.stream(gs):
            if ready is not None:
                gs.wait_event(ready)
            else:
                gs.wait_stream(main)
            if os.environ.get('RD_GEOM_COMPOSITE', '1') != '0':
                # two library calls per branch around the read (rd_geometry_begin / rd_geometry_finish), one allocation each
                scal = torch.empty(5 * len(vfes), dtype=torch.int32, device=dev)
                begun = [v.prelude_begin(batch_dict, 3, scal[5 * i:5 * i + 5]) for i, v in enumerate(vfes)]
                vals = scal.tolist()                                                             # the only read (geometry stream only)
                levels = [v.prelude_finish(batch_dict, st, vals[5 * i:5 * i + 5]) for i, (v, st) in enumerate(zip(vfes, begun))]
                shared = [(st[0], st[1][0], lvl.coords) for st, lvl in zip(begun, levels)]       # points + the two allocations
            else:
                begun = [v.geometry_begin(batch_dict) for v in vfes]
                # every level's rank grid is marked from the grid above it with static launch shapes, so ALL sizes of both branches
                # (pillars, in-range points, 3 pyramid levels) come back in ONE read
                marked = [SP.mark_pyramid(st[1], True, st[3], v.grid_y, v.grid_x, 3) for v, (st, _) in zip(vfes, begun)]
                scalars = [t for (_, sc), mk in zip(begun, marked) for t in (*sc, *[m[3].long() for m in mk])]
                vals = torch.stack(scalars).tolist()                                             # the only read (geometry stream only)
                levels, shared = [], []
                for i, (v, (st, _), mk) in enumerate(zip(vfes, begun, marked)):
                    base = 5 * i
                    lvl = v.geometry_finish(batch_dict, st, int(vals[base]), int(vals[base + 1]))
                    SP.finish_pyramid(lvl, mk, vals[base + 2:base + 5])
                    levels.append(lvl)
                    shared.append((st[0], st[2], *lvl.tensors()))
            done = torch.cuda.Event()
            done.record(gs)
        main.wait_event(done)
        # tensors allocated on the geometry stream are read on the main / teacher streams: tell the caching allocator
        # (record_stream acts on the allocation, so one view per allocation is enough)
        streams = [main]
        if getattr(self, '_teacher_stream', None) is None and self.no_grad_module:
            self._teacher_stream = torch.cuda.Stream(dev, priority=int(os.environ.get('RD_TEACHER_PRIO', '0')))
        if getattr(self, '_teacher_stream', None) is not None:
            streams.append(self._teacher_stream)
        for tensors in shared:
            tensors[0].record_stream(gs)                                # the points were read on the geometry stream
            for t in tensors:                                           # (they may be a converted copy made on that stream)
                for s_ in streams:
                    t.record_stream(s_)

    def post_processing(self, batch_dict):
        """pillarnet.py:80-96: the head already decoded + NMS-ed (`final_box_dicts`); add the recall records."""
        post_process_cfg = self.model_cfg.POST_PROCESSING
        final_pred_dict = batch_dict['final_box_dicts']
        recall_dict = {}
        for index in range(batch_dict['batch_size']):
            recall_dict = self.generate_recall_record(box_preds=final_pred_dict[index]['pred_boxes'], recall_dict=recall_dict,
                                                      batch_index=index, data_dict=batch_dict, thresh_list=post_process_cfg.RECALL_THRESH_LIST)
        return final_pred_dict, recall_dict

    def get_training_loss(self):
        loss_rpn, tb_dict = self.dense_head.get_loss()
        tb_dict = {'loss_rpn': loss_rpn.detach(), **tb_dict}
        return loss_rpn, tb_dict, {}

    def get_training_distll_loss(self, batch_dict):
        loss_feature, tb_dict = self.radar_backbone_2d.get_loss(batch_dict)
        loss_rpn, _tb_dict = self.radar_dense_head.get_loss()
        tb_dict.update(_tb_dict)
        return loss_feature + loss_rpn, tb_dict, {}

    def get_training_wo_distll_loss(self, batch_dict):
        loss_rpn, tb_dict = self.radar_dense_head.get_loss()
        return loss_rpn, tb_dict, {}
