"""Optimizer step + training-step driver of the hot path.

Semantics of the reference loop (tools/train_utils/train_utils.py:44-64, optimization/__init__.py:19-54,
fastai_optim.py:135-152, learning_schedules_fastai.py:44-77): OneCycle(lr_max, moms, div_factor, pct_start) sets lr and
beta1 per iteration; zero_grad; forward; backward (DDP all-reduce overlapped); clip_grad_norm_(10); decoupled weight decay
p *= 1 - wd*lr on EVERY trainable parameter (bn_wd=True); Adam(betas=(mom, 0.99), eps 1e-8).

MI355X design: the ~500 parameter tensors are described by one device table; gradient norm and decay+Adam are two
multi-tensor HIP launches (optim.hip), the clip coefficient never visits the host, Adam state lives in two flat buffers.
"""
import ctypes
import math
import weakref

import numpy as np
import torch

from . import autograd as A
from . import native
from .kernels import _p, _stream
from .native import check


def annealing_cos(start, end, pct):
    return end + (start - end) / 2 * (math.cos(math.pi * pct) + 1)


class OneCycle:
    """lr / momentum phases of learning_schedules_fastai.py:60-77 (+ LRSchedulerStep.step :44-50)."""

    def __init__(self, optimizer, total_step, lr_max, moms, div_factor, pct_start):
        self.optimizer, self.total_step = optimizer, int(total_step)
        self.lr_max, self.moms, self.div_factor, self.pct_start = lr_max, list(moms), div_factor, pct_start
        low = lr_max / div_factor
        p1 = int(self.total_step * pct_start)
        self.lr_phases = [(0, p1, (low, lr_max)), (p1, self.total_step, (lr_max, low / 1e4))]
        self.mom_phases = [(0, p1, (self.moms[0], self.moms[1])), (p1, self.total_step, (self.moms[1], self.moms[0]))]
        optimizer.lr, optimizer.mom = low, self.moms[0]

    def step(self, step, epoch=None):
        for s, e, (a, b) in self.lr_phases:
            if step >= s:
                self.optimizer.lr = annealing_cos(a, b, (step - s) / (e - s))
        for s, e, (a, b) in self.mom_phases:
            if step >= s:
                self.optimizer.mom = annealing_cos(a, b, (step - s) / (e - s))


class _OptTensor(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("numel", ctypes.c_int64)]


class FusedAdamOneCycle:
    """`adam_onecycle` optimizer of the reference (OptimWrapper over Adam with true_wd, bn_wd) as fused HIP launches."""

    def __init__(self, params, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, wd=0.01, grad_clip=10.0, ref_groups=None):
        """ref_groups: the reference optimizer's numbering of these parameters -- a list of lists of indices into the trainable
        `params` (build_optimizer derives the two groups of split_bn_bias); only state_dict()/load_state_dict() use it."""
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise RuntimeError("FusedAdamOneCycle needs contiguous fp32 parameters on one device")
        self.lr, self.mom, self.beta2, self.eps, self.wd, self.grad_clip = lr, betas[0], betas[1], eps, wd, grad_clip
        self.step_count = 0
        n = sum(p.numel() for p in self.params)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.offsets = np.cumsum([0] + [p.numel() for p in self.params])
        chunk = native.lib().rd_opt_chunk_elems()
        chunks = []
        for i, p in enumerate(self.params):
            for off in range(0, p.numel(), chunk):
                chunks.append((i, off))
        self.n_chunks = len(chunks)
        self.chunks_dev = torch.tensor(chunks, dtype=torch.int32, device=dev).contiguous()
        # Descriptor tables go host -> device asynchronously every step (gradient tensors are new allocations each step).  The host
        # may run a whole step ahead of the GPU, so the pinned staging buffer is a ring: a slot is rewritten only
        # after the copy that read it has completed (event), never while a DMA may still be reading it.
        nbytes = len(self.params) * ctypes.sizeof(_OptTensor)
        self._ring = 4
        self._slot = 0
        self.table_host = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() if dev.type == "cuda" else torch.empty(nbytes, dtype=torch.uint8)
                           for _ in range(self._ring)]
        self.table_dev = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(self._ring)]
        self._copied = [None] * self._ring
        self._slot_sig = [None] * self._ring
        self.norm_out = torch.zeros(2, dtype=torch.float32, device=dev)
        self.ws = torch.empty(self.n_chunks, dtype=torch.float32, device=dev)
        self.ref_groups = [list(g) for g in ref_groups] if ref_groups is not None else [list(range(len(self.params))), []]
        if sorted(i for g in self.ref_groups for i in g) != list(range(len(self.params))):
            raise ValueError("ref_groups must number every trainable parameter exactly once")
        # torch.optim.Adam keeps a step count PER PARAMETER and skips parameters whose gradient is None; skipped_dev[i] = optimizer
        # steps parameter i sat out.  Device-owned (rd_adam_step counts a sat-out step itself): under data parallelism "sat out" is a
        # group-wide fact (the all-reduced presence mask) that the host never reads; `skipped` reads the counters back on demand
        self.skipped_dev = torch.zeros(len(self.params), dtype=torch.int32, device=dev)
        # steps skipped as a whole by loss-scaling overflow (AmpScaler): not Adam steps for any parameter; counted on the device by
        # rd_grad_norm, owned here so that state_dict() / load_state_dict() keep every parameter's Adam `step` free of them
        self.overflows = torch.zeros(1, dtype=torch.int32, device=dev)
        self._amp_used = False
        self.present = None            # data parallelism: per-parameter "some rank has a gradient" mask, all-reduced every step
        self._static_cols = None
        self._param_ptrs = None
        self._grad_col = None
        self.flat_grad = None          # data parallelism: see enable_flat_allreduce()
        self.process_group = None

    @property
    def skipped(self):
        """Per-parameter count of optimizer steps sat out (numpy; reads the device counters: checkpoint / test time only)."""
        return self.skipped_dev.cpu().numpy()

    def enable_flat_allreduce(self, process_group=None, bucket_mb=None, overlap=None):
        """Data-parallel gradient exchange without DistributedDataParallel: the ~500 gradient tensors are packed into a flat fp32
        buffer (laid out like the Adam moments), summed over the ranks (RCCL over xGMI) and the norm / Adam kernels read the averaged
        gradients straight from that buffer (flat_grad, grad_scale = 1 / world).  Parameters must start equal on all ranks
        (dist.broadcast_parameters).
        Default (RD_DDP_OVERLAP unset or 0): ONE pack launch + ONE all-reduce of the whole buffer after backward -- 100 MB over xGMI,
        0.5-1 ms on the critical path of a 17.4 ms step, and no host work during backward.
        overlap (RD_DDP_OVERLAP=1): the buffer is cut into ~bucket_mb (25) MB buckets in reverse parameter order; a
        post-accumulate-grad hook per parameter counts a bucket down and, when its last gradient exists, packs the bucket (one
        launch, rd_pack_grads_list) and starts its all-reduce on a communication stream that first waits for the main and the
        weight-gradient streams -- the head's 7 MB travel while DenseEnc / CMA / SparseEnc are still in their backward; step()
        only waits for the collectives.  OFF by default since round 3: rehearsed on the GPU in a world of one rank (dist.rehearsal:
        every hook, bucket launch and RCCL call executes, the collectives move nothing) the overlapped path costs the host-bound step
        17.4 -> 19.1 ms with the loop on torch's default stream (265 Python hook calls in the autograd thread + ~0.3 ms per bucket
        launch) and 24.1 ms with the loop on its high-priority stream (the communication stream's waits for the weight-gradient
        stream then hold up packets of the stream that shares its hardware queue) -- more than the all-reduce it hides."""
        import os
        from .dist import GradBuckets
        n = int(self.offsets[-1])
        # the presence mask (one float per parameter, see _presence) lives behind the gradients in the same allocation: without
        # buckets ONE collective per step carries both
        self._flat_all = torch.zeros(n + len(self.params), dtype=torch.float32, device=self.params[0].device)
        self.flat_grad = self._flat_all[:n]
        self.present = self._flat_all[n:]
        self._presence_fresh = False          # the mask was summed together with the gradients of THIS exchange
        self.process_group = process_group
        if overlap is None:
            overlap = os.environ.get("RD_DDP_OVERLAP", "0") != "0"
        self.buckets = None
        if overlap:
            mb = float(bucket_mb if bucket_mb is not None else os.environ.get("RD_DDP_BUCKET_MB", "25"))
            self.buckets = GradBuckets([p.numel() for p in self.params], int(mb * (1 << 20)))
            self._works = []
            self._work_of = {}
            self._pack_cache = {}
            # high priority, like the training loop's own stream.  (Normal priority was measured worse still: 32 ms per step in the
            # one-rank rehearsal against 24 ms -- whichever hardware queue the communication stream shares, its waits for the
            # weight-gradient stream hold the packets queued behind them.)
            self._comm_stream = torch.cuda.Stream(self.params[0].device, priority=-1) if self.params[0].is_cuda else None
            # (a bucket is packed mid-backward: _launch_bucket first runs the deferred weight-gradient re-layouts that are due)
            for i, p in enumerate(self.params):
                p.register_post_accumulate_grad_hook(lambda _p, i=i: self._grad_ready(i))
            # gradients delivered outside autograd's AccumulateGrad (autograd.ConcatLeaves) announce themselves here
            index = {id(p): i for i, p in enumerate(self.params)}
            me = weakref.ref(self)

            def _delivered(leaves):
                opt = me()
                if opt is None:
                    return
                for leaf in leaves:
                    i = index.get(id(leaf))
                    if i is not None and opt.params[i] is leaf:
                        b = opt.buckets.ready(i)
                        if b is not None:
                            opt._launch_bucket(b)

            A.GRAD_LISTENERS[:] = [cb for cb in A.GRAD_LISTENERS if getattr(cb, '_owner', lambda: None)() is not None]
            _delivered._owner = me
            A.GRAD_LISTENERS.append(_delivered)

    def _grad_ready(self, i):
        b = self.buckets.ready(i)
        if b is not None:
            self._launch_bucket(b)

    def _launch_bucket(self, b):
        """Pack bucket b's gradients into its slice of the flat buffer and start the slice's all-reduce."""
        import torch.distributed as dist
        from .native import PackJob
        lo, hi = self.buckets.ranges[b]
        e0, e1 = int(self.offsets[lo]), int(self.offsets[hi])
        dev = self.params[0].device
        comm = self._comm_stream
        main = torch.cuda.current_stream(dev)
        ev = torch.cuda.Event()
        ev.record(main)
        comm.wait_event(ev)
        side = A._WGRAD_STREAMS.get(dev)
        if side is not None:
            if A._DEFERRED_LAYOUT:                 # the re-layouts of this bucket's (and every other accumulated) gradients: one launch
                A._set_stream(side)
                try:
                    A._flush_deferred_layouts(only_accumulated=True)
                finally:
                    A._set_stream(main)
            comm.wait_stream(side)                 # weight gradients are produced on the side stream
        base = self.flat_grad.data_ptr()
        prev = self._work_of.pop(b, None)
        # The pack descriptors of a bucket only depend on where its gradients live, and in steady state the allocator hands every
        # step's gradients the same addresses: the ctypes arrays are rebuilt (and the gradients re-validated) only when that
        # signature changes, or every 64th launch (filling ~480 descriptors cost 1.3 ms of host time per step, round 3).
        grads = [p.grad for p in self.params[lo:hi]]
        sig = tuple([g.data_ptr() if g is not None else 0 for g in grads])
        cache = self._pack_cache.get(b)
        if cache is None or cache[0] != sig or cache[2] >= 64:
            jobs = []
            for s in range(lo, hi, 128):
                part = range(s, min(hi, s + 128))
                arr = (PackJob * len(part))()
                for k, i in enumerate(part):
                    g = self.params[i].grad
                    if g is not None and (not g.is_contiguous() or g.dtype != torch.float32):
                        g = self.params[i].grad = g.float().contiguous()
                    arr[k].src = g.data_ptr() if g is not None else None
                    arr[k].dst = base + 4 * int(self.offsets[i])
                    arr[k].numel = self.params[i].numel()
                jobs.append((arr, len(part)))
            sig = tuple([(p.grad.data_ptr() if p.grad is not None else 0) for p in self.params[lo:hi]])
            cache = self._pack_cache[b] = [sig, jobs, 0, base]
        cache[2] += 1
        with torch.cuda.stream(comm):
            if prev is not None:
                prev.wait()                        # a dirty bucket is re-packed: its first collective must have left the slice
            for arr, n_jobs in cache[1]:
                check(native.lib().rd_pack_grads_list(arr, n_jobs, _stream()), "rd_pack_grads_list")
            w = dist.all_reduce(self.flat_grad[e0:e1], op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)
            self._works.append(w)
            self._work_of[b] = w

    def zero_grad(self):
        native.lib().clear_grads(self.params)          # p.grad = None for every parameter (one loop in C)
        if getattr(self, 'buckets', None) is not None and (self._works or any(self.buckets.seen)):
            # a backward pass that was not followed by step() (it raised, or its step was abandoned): collectives in flight are
            # drained and the bucket bookkeeping starts clean, instead of carrying half-counted buckets into the next pass
            for w in self._works:
                w.wait()
            self._works, self._work_of = [], {}
            self.buckets.reset()

    def _fill_table(self):
        # gradient tensors are new objects every step but, in steady state, the allocator hands back the same addresses: when a
        # ring slot was filled from exactly these pointers its device copy is still valid -- no refill, no host->device copy
        # (the loop over the ~480 parameters -- p.grad, contiguity, dtype, data_ptr -- runs in the C helper native.lib().grad_ptrs)
        cur = self._grad_col
        if cur is None or cur.shape[0] != len(self.params):
            cur = self._grad_col = np.zeros(len(self.params), dtype=np.int64)
        L = native.lib()
        while True:
            r = L.grad_ptrs(self.params, torch.float32, cur)          # 0 for a parameter unused this step (Adam skips it: decay only)
            if r > -2:
                break
            bad = self.params[-2 - r]
            bad.grad = bad.grad.float().contiguous()
        pp = self._param_ptrs          # parameters keep their storage (optimizers update in place): revalidated by the two ends
        if pp is None or pp[0] != self.params[0].data_ptr() or pp[-1] != self.params[-1].data_ptr():
            pp = self._param_ptrs = tuple(p.data_ptr() for p in self.params)
        for slot in range(self._ring):
            sg = self._slot_sig[slot]
            if sg is not None and sg[1] is pp and np.array_equal(sg[0], cur):
                return self.table_dev[slot]
        grads = cur
        self._slot = (self._slot + 1) % self._ring
        ev = self._copied[self._slot]
        if ev is not None:
            ev.synchronize()                   # the copy issued `ring` steps ago out of this slot (and the kernels that read the
        host, dev_t = self.table_host[self._slot], self.table_dev[self._slot]     # device copy, stream-ordered before it) are done
        # the table is 5 int64 columns per parameter; only the gradient column changes from step to step (a Python loop writing
        # ctypes fields cost 1 ms per step for ~480 parameters)
        tab = host.numpy().view(np.int64).reshape(len(self.params), 5)
        static = self._static_cols
        if static is None or static[0] != pp:
            m0, v0 = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
            cols = np.empty((len(self.params), 5), dtype=np.int64)
            cols[:, 0] = [p.data_ptr() for p in self.params]
            cols[:, 2] = m0 + 4 * self.offsets[:-1]
            cols[:, 3] = v0 + 4 * self.offsets[:-1]
            cols[:, 4] = [p.numel() for p in self.params]
            static = self._static_cols = (tuple(int(v) for v in cols[:, 0]), cols)
        tab[:] = static[1]
        tab[:, 1] = grads
        self._slot_sig[self._slot] = (cur.copy(), pp)
        dev_t.copy_(host, non_blocking=True)
        if dev_t.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self._copied[self._slot] = ev
        return dev_t

    def allreduce_gradients(self, table=None):
        """Flat-buffer mode: pack every p.grad into self.flat_grad (one launch) and sum it over the ranks (one collective).
        Returns (flat buffer, scale = 1 / world size), or (None, 1.0) when data parallelism is off / handled by DDP."""
        if self.flat_grad is None:
            return None, 1.0
        import torch.distributed as dist
        if getattr(self, 'buckets', None) is not None:
            for b in self.buckets.open_buckets():          # parameters that received no gradient keep their bucket open until here
                self._launch_bucket(b)
            for w in self._works:
                w.wait()                                   # the current stream waits for the collective
            torch.cuda.current_stream(self.flat_grad.device).wait_stream(self._comm_stream)
            self._works, self._work_of = [], {}
            self.buckets.reset()
            return self.flat_grad, 1.0 / dist.get_world_size(self.process_group)
        if table is None:
            table = self._fill_table()
        check(native.lib().rd_pack_grads(_p(table), _p(self.chunks_dev), self.n_chunks, _p(self.flat_grad), _stream()), "rd_pack_grads")
        check(native.lib().rd_grad_presence(_p(table), len(self.params), _p(self.present), _stream()), "rd_grad_presence")
        dist.all_reduce(self._flat_all, op=dist.ReduceOp.SUM, group=self.process_group)          # gradients + presence counts: one collective
        self._presence_fresh = True
        return self.flat_grad, 1.0 / dist.get_world_size(self.process_group)

    def _presence(self, table):
        """Flat-buffer mode: the group-wide "has a gradient" mask of this step (rd_grad_presence + one MAX all-reduce of ~500 floats).
        A parameter without a gradient on THIS rank only (a branch its batch did not use) then takes part like on the other ranks,
        with the averaged gradient from the flat buffer -- same norm, same clip coefficient, same Adam step count everywhere."""
        import torch.distributed as dist
        if self._presence_fresh:          # summed together with the gradients (allreduce_gradients without buckets)
            self._presence_fresh = False
            return self.present
        check(native.lib().rd_grad_presence(_p(table), len(self.params), _p(self.present), _stream()), "rd_grad_presence")
        dist.all_reduce(self.present, op=dist.ReduceOp.MAX, group=self.process_group)
        return self.present

    def step(self, inv_loss_scale=None, overflow_count=None):
        """clip_grad_norm_(grad_clip) + OptimWrapper.step(); returns the device tensor [total_norm, clip_coef].
        inv_loss_scale: device scalar 1 / S when the loss was multiplied by S before backward (AmpScaler): norm and update use g / S
        and a non-finite norm skips the whole update on the device, as GradScaler.unscale_ / step do (train_utils.py:60-64);
        such a step is counted in self.overflows (they are not Adam steps: bias correction uses step - overflows).
        overflow_count: accepted for callers of the round-2 signature; the optimizer's own counter is what is used."""
        table = self._fill_table()
        L = native.lib()
        flat, scale = self.allreduce_gradients(table)
        present = self._presence(table) if flat is not None else None
        clip = None
        amp = inv_loss_scale is not None
        overflow_count = self.overflows if (amp or self._amp_used) else None
        self._amp_used = self._amp_used or amp
        if amp or (self.grad_clip is not None and self.grad_clip > 0):
            max_norm = float(self.grad_clip) if (self.grad_clip is not None and self.grad_clip > 0) else 0.0
            check(L.rd_grad_norm(_p(table), _p(self.chunks_dev), self.n_chunks, max_norm, _p(self.norm_out),
                                 _p(self.ws), self.ws.numel() * 4, _p(flat), scale, _p(inv_loss_scale), _p(overflow_count if amp else None),
                                 _p(present), _stream()), "rd_grad_norm")
            clip = self.norm_out
        self.step_count += 1
        check(L.rd_adam_step(_p(table), _p(self.chunks_dev), self.n_chunks, float(self.lr), float(self.mom), float(self.beta2),
                             float(self.eps), float(self.wd), self.step_count, _p(self.skipped_dev), _p(clip), _p(flat), scale,
                             _p(inv_loss_scale), int(amp), _p(overflow_count), _p(present), _stream()), "rd_adam_step")
        A.bump_weights_epoch()                 # parameters changed through raw pointers: invalidate cached weight layouts
        return self.norm_out

    def state_dict(self):
        """The inner torch.optim.Adam's state_dict as the reference stores it under `optimizer_state` (OptimWrapper passes
        state_dict through to `self.opt`, fastai_optim.py:162-164; written by train_utils.py:253-270): per-parameter
        {step, exp_avg, exp_avg_sq} keyed by the reference's parameter numbering (group 0: non-BatchNorm leaves, group 1: BatchNorm
        leaves, split_bn_bias fastai_optim.py:16-28), parameters that never had a gradient carry no state."""
        state, off = {}, self.offsets
        order = [i for g in self.ref_groups for i in g]
        skipped = self.skipped
        overflows = int(self.overflows.item()) if self._amp_used else 0          # steps GradScaler skipped are nobody's Adam step
        for idx, i in enumerate(order):
            own = self.step_count - int(skipped[i]) - overflows
            if own <= 0:
                continue
            p = self.params[i]
            state[idx] = {"step": own, "exp_avg": self.exp_avg[off[i]:off[i + 1]].view(p.shape).clone(),
                          "exp_avg_sq": self.exp_avg_sq[off[i]:off[i + 1]].view(p.shape).clone()}
        groups, base = [], 0
        for g in self.ref_groups:
            groups.append({"lr": self.lr, "betas": (self.mom, self.beta2), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                           "params": list(range(base, base + len(g)))})
            base += len(g)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        """Accepts the layout above -- i.e. a reference checkpoint's `optimizer_state` -- and the flat dictionary this class wrote
        in round 1 ({step, exp_avg, exp_avg_sq, lr, mom})."""
        if "state" in sd and "param_groups" in sd:
            sizes = [len(g["params"]) for g in sd["param_groups"]]
            if sizes != [len(g) for g in self.ref_groups]:
                raise RuntimeError(f"optimizer state has parameter groups of sizes {sizes}, this model has "
                                   f"{[len(g) for g in self.ref_groups]} (non-BatchNorm / BatchNorm trainable parameters)")
            order = [i for g in self.ref_groups for i in g]
            steps = np.zeros(len(self.params), dtype=np.int64)
            self.exp_avg.zero_(); self.exp_avg_sq.zero_()
            for idx, st in sd["state"].items():
                i = order[int(idx)]
                p = self.params[i]
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise RuntimeError(f"optimizer state {idx}: moment shape {tuple(st['exp_avg'].shape)} != parameter shape {tuple(p.shape)}")
                steps[i] = int(st["step"])
                self.exp_avg[self.offsets[i]:self.offsets[i + 1]].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[self.offsets[i]:self.offsets[i + 1]].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = int(steps.max()) if len(steps) else 0
            self.skipped_dev.copy_(torch.from_numpy((self.step_count - steps).astype(np.int32)))
            self.overflows.zero_()             # the stored `step`s are free of overflow-skipped steps: counting restarts consistently
            g0 = sd["param_groups"][0]
            self.lr, self.mom = float(g0["lr"]), float(g0["betas"][0])
            return
        if not {"step", "exp_avg", "exp_avg_sq"} <= set(sd):
            raise RuntimeError("unrecognised optimizer state: expected torch.optim.Adam's {state, param_groups} or this build's flat "
                               "{step, exp_avg, exp_avg_sq, lr, mom}")
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.lr, self.mom = sd["lr"], sd["mom"]
        self.skipped_dev.zero_()
        self.overflows.zero_()


def reference_param_groups(model):
    """The reference optimizer's parameter numbering (optimization/__init__.py:19-33 + fastai_optim.py:16-28,93-96,117-123): the
    model is flattened to its leaf modules in definition order, BatchNorm leaves are split off into a second group, and each group
    lists its trainable parameters.  Returns (trainable parameters in model.parameters() order, [group0 indices, group1 indices])."""
    bn_types = (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d, torch.nn.SyncBatchNorm)
    params = [p for p in model.parameters() if p.requires_grad]
    index = {id(p): i for i, p in enumerate(params)}
    groups, seen = ([], []), set()
    for leaf in (m for m in model.modules() if not any(True for _ in m.children())):
        dst = groups[1] if isinstance(leaf, bn_types) else groups[0]
        for p in leaf.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                dst.append(index[id(p)])
    missing = [i for i in range(len(params)) if i not in set(groups[0]) | set(groups[1])]
    groups[0].extend(missing)           # parameters owned by non-leaf modules (none in this model): the reference never optimises them
    return params, [groups[0], groups[1]]


def convert_sync_batchnorm(model, process_group=None):
    """--sync_bn of the reference (tools/train.py:34,144-145): torch.nn.SyncBatchNorm.convert_sync_batchnorm(model) swaps every
    BatchNorm container for a SyncBatchNorm one (same parameters, buffers and state_dict names).  The arithmetic stays in norm.hip /
    vfe.hip: a train-mode layer all-reduces its (sum, sum of squares, count) in the forward and its (grad_gamma, grad_beta) in the
    backward over the process group -- two small collectives per layer, values handed to the kernels in device memory -- and
    normalises with the group-wide statistics (rd_bn_train_fwd_sync, rd_bn_bwd_reduce / rd_bn_bwd_apply, rd_vfe_backward_*).
    Eval-mode layers (the frozen teacher) are untouched, as in torch."""
    from . import autograd as A
    model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model, process_group)
    A.SYNC_BN[0] = True          # also covers layers whose containers are batched on the channel axis (CenterHead branches)
    return model


def build_optimizer(model, optim_cfg):
    if optim_cfg.OPTIMIZER != 'adam_onecycle':
        raise NotImplementedError("the distill config trains with adam_onecycle")
    betas = tuple(optim_cfg.get('BETAS', (0.9, 0.99)))
    params, groups = reference_param_groups(model)
    return FusedAdamOneCycle(params, lr=3e-3, betas=betas, wd=optim_cfg.WEIGHT_DECAY, grad_clip=optim_cfg.GRAD_NORM_CLIP, ref_groups=groups)


def build_scheduler(optimizer, total_iters_each_epoch, total_epochs, last_epoch, optim_cfg):
    total_steps = total_iters_each_epoch * total_epochs
    return OneCycle(optimizer, total_steps, optim_cfg.LR, list(optim_cfg.MOMS), optim_cfg.DIV_FACTOR, optim_cfg.PCT_START), None


class AmpScaler:
    """Dynamic loss scaling of the reference's `--use_amp` path (torch.cuda.amp.GradScaler(init_scale=LOSS_SCALE_FP16 or 2**16),
    train_utils.py:23,57-64; torch defaults growth 2.0, backoff 0.5, growth interval 2000) without host synchronisation: the scale
    and the growth counter live on the device; unscale_ + clip + step are the fused optimizer's own launches (g / S inside the norm
    and Adam kernels, a non-finite norm skips the update); update() is four tiny device ops.
    The arithmetic under --use_amp is `autocast()` below: the convolution / linear products of forward AND backward run on operands
    rounded to bf16 (one MFMA term instead of bf16x3's three) with fp32 accumulation; activations, gradients and master weights stay
    fp32 in memory (torch's autocast stores half-precision activations: storage here is wider, the products are the same class)."""

    def __init__(self, device, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self.enabled = bool(enabled)
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self._scale = torch.full((), float(init_scale), dtype=torch.float32, device=device)
        self._inv = torch.full((1,), 1.0 / float(init_scale), dtype=torch.float32, device=device)
        self._tracker = torch.zeros((), dtype=torch.int32, device=device)

    def get_scale(self):
        return float(self._scale) if self.enabled else 1.0

    def scale(self, loss):
        return loss * self._scale if self.enabled else loss

    def step(self, optimizer):
        """unscale_ + clip_grad_norm_ + optimizer.step(): one pass over the gradients; returns [norm of g / S, clip coefficient]."""
        if not self.enabled:
            return optimizer.step()
        self._norm = optimizer.step(inv_loss_scale=self._inv)          # overflow-skipped steps are counted by the optimizer (state_dict)
        return self._norm

    def update(self):
        if not self.enabled:
            return
        found_inf = ~torch.isfinite(self._norm[0])
        grow = (~found_inf) & (self._tracker + 1 >= self.growth_interval)
        self._scale = torch.where(found_inf, self._scale * self.backoff_factor, torch.where(grow, self._scale * self.growth_factor, self._scale))
        self._tracker = torch.where(found_inf | grow, torch.zeros_like(self._tracker), self._tracker + 1)
        self._inv.copy_((1.0 / self._scale).reshape(1))


class autocast:
    """`with autocast(enabled):` -- the arithmetic of the reference's `torch.cuda.amp.autocast(enabled=use_amp)` block
    (tools/train_utils/train_utils.py:57-58) for this build's kernels: MFMA convolutions / GEMMs take bf16-rounded operands
    (rd_set_conv_math(1) + rd_set_mfma_terms(1)), fp32 accumulate.  Unlike torch's context it must also cover `backward()`: the data and
    weight gradient kernels read the switch when they are launched.  Restores the previous modes on exit."""

    def __init__(self, enabled=True):
        self.enabled = bool(enabled)

    def __enter__(self):
        from . import kernels as K
        if self.enabled:
            self._prev = (K.get_conv_math(), K.get_mfma_terms())
            K.set_conv_math("bf16x3")
            K.set_mfma_terms(1)
        return self

    def __exit__(self, *exc):
        from . import kernels as K
        if self.enabled:
            K.set_conv_math(self._prev[0])
            K.set_mfma_terms(self._prev[1])
        return False


def use_training_stream(device, priority=-1):
    """Make a HIGH-priority HIP stream the current stream of `device` for the training loop (call once, before the first step) and
    return it.  The step's critical path -- student forward, BatchNorm backward passes, data gradients -- then has its workgroups
    dispatched ahead of the teacher's and the weight-gradient stream's whenever both wait for a CU: +1 % (18.6 -> 18.4 ms per step,
    measured in alternation).  priority 0 keeps torch's default stream."""
    if device.type != "cuda" or not priority:
        return None
    s = torch.cuda.Stream(device, priority=priority)
    s.wait_stream(torch.cuda.current_stream(device))          # parameter uploads / initialisation enqueued so far
    torch.cuda.set_stream(s)
    return s


def train_step(model, optimizer, lr_scheduler, model_func, batch, accumulated_iter, scaler=None):
    """One iteration of train_one_epoch (train_utils.py:44-64) without logging: returns (loss tensor, tb_dict).
    scaler: an AmpScaler for the reference's `--use_amp` loop (scale the loss, unscale + clip + step in one pass, update)."""
    lr_scheduler.step(accumulated_iter)
    model.train()
    optimizer.zero_grad()
    amp = scaler is not None and scaler.enabled
    with autocast(enabled=amp):          # (forward and backward: see autocast)
        loss, tb_dict, disp_dict = model_func(model, batch)
        if amp:
            scaler.scale(loss).backward()
        else:
            loss.backward()
    if amp:
        scaler.step(optimizer)
        scaler.update()
    else:
        optimizer.step()
    return loss, tb_dict


# ---------------------------------------------------------------------------------------------- checkpoints (train_utils.py:253-293)
def model_state_to_cpu(model_state):
    return type(model_state)((k, v.cpu()) for k, v in model_state.items())


def checkpoint_state(model=None, optimizer=None, epoch=None, it=None):
    """Same dictionary layout as the reference (`epoch`, `it`, `model_state`, `optimizer_state`, `version`; train_utils.py:253-270):
    `model_state` names equal the reference's and `optimizer_state` is torch.optim.Adam's own {state, param_groups} in the
    reference's parameter numbering (FusedAdamOneCycle.state_dict), which FusedAdamOneCycle.load_state_dict also reads back from a
    reference checkpoint."""
    optim_state = optimizer.state_dict() if optimizer is not None else None
    model_state = None
    if model is not None:
        m = model.module if isinstance(model, torch.nn.parallel.DistributedDataParallel) else model
        model_state = model_state_to_cpu(m.state_dict())
    return {'epoch': epoch, 'it': it, 'model_state': model_state, 'optimizer_state': optim_state, 'version': 'radardistill_amd'}


def save_checkpoint(state, filename='checkpoint'):
    torch.save(state, '{}.pth'.format(filename))
