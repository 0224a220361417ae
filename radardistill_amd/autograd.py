"""torch.autograd glue around the HIP kernels (host side stays Python on PyTorch-ROCm, as BASELINE.json's
north_star asks).  Each Function's forward and backward run hand-written kernels through the C ABI; torch only
owns the tensors and the graph.
"""
import os
import weakref

import torch

from . import kernels as K

_DEBUG = bool(int(os.environ.get("RD_DEBUG_CHECK", "0")))     # developer aid: re-derive backward results with torch ops and report mismatches


def _dbg_report(name, got, ref):
    err = float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
    if err > 1e-4:
        d = (got - ref).abs().max(1)[0]
        rows = torch.topk(d, min(5, d.numel())).indices.tolist()
        print(f"[RD_DEBUG] {name}: rel err {err:.3e} worst rows {rows} of {got.shape}", flush=True)

# bumped by the optimizer after every parameter update done through raw pointers (tensor._version does not see those)
_WEIGHTS_EPOCH = [0]
_LAYOUT_CACHE = {}


_FROZEN_EPOCH = [-1]          # frozen parameters (the teacher) only change through torch (`_version`) -- or through a raw write that says so


def bump_weights_epoch(frozen=False):
    """frozen=True: frozen parameters / buffers were written behind torch's back as well (dist.broadcast_parameters copies through
    `.data`): their cached layouts, split operands and folded BatchNorm coefficients are invalidated too."""
    _WEIGHTS_EPOCH[0] += 1
    if frozen:
        _FROZEN_EPOCH[0] -= 1


class _ZeroArena:
    """Per-step pool of zero-initialised fp32 scratch (BatchNorm statistics, weight-gradient accumulators): ONE memset per
    step instead of one torch.zeros launch per layer.  `begin_step()` re-zeroes and rewinds; slices stay valid until the next
    begin_step() (all consumers finish inside the step they were handed out in, on the same stream)."""

    def __init__(self):
        self.buf = None
        self.off = 0
        self.high = 0

    def begin_step(self, device):
        need = max(self.high, 1 << 20)
        if self.buf is None or self.buf.device != device or self.buf.numel() < need:
            self.buf = torch.zeros(int(need * 1.25), dtype=torch.float32, device=device)
        else:
            self.buf[:max(self.off, 1)].zero_()
        self.off = 0

    def take(self, n, device):
        n_al = (n + 63) // 64 * 64
        if self.buf is None or self.buf.device != device or self.off + n_al > self.buf.numel():
            self.high = max(self.high, self.off + n_al)          # grow on the next step; this request falls back to a fresh tensor
            self.off += n_al
            return torch.zeros(n, dtype=torch.float32, device=device)
        out = self.buf[self.off:self.off + n]
        self.off += n_al
        self.high = max(self.high, self.off)
        return out


class _GradArena:
    """Zero-filled accumulators that may ESCAPE as parameter gradients (weight / bias / BatchNorm gradients are accumulated with
    atomics and handed to autograd as they are).  One fresh torch.zeros per step, sized by the previous step's demand; the arena
    drops its reference at the next begin_step(), so the buffer lives exactly as long as some .grad still aliases it (gradient
    accumulation over several backward passes stays correct)."""

    def __init__(self):
        self.buf = None
        self.off = 0
        self.high = 0
        self.step_high = 0

    def begin_step(self, device=None):
        self.high = max(self.high, self.step_high)
        self.buf, self.off, self.step_high = None, 0, 0
        if self.high > 0 and device is not None and device.type == "cuda":
            # allocated and zero-filled HERE, on the forward's stream: slices are handed out later from backward code that may run
            # under a side stream (weight gradients), and the fill must not be ordered on that stream
            self.buf = torch.zeros(int(self.high * 1.05) + 4096, dtype=torch.float32, device=device)

    def take(self, n, device):
        n_al = (n + 63) // 64 * 64
        self.step_high += n_al
        if self.buf is None or self.buf.device != device or self.off + n_al > self.buf.numel():
            return torch.zeros(n, dtype=torch.float32, device=device)
        out = self.buf[self.off:self.off + n]
        self.off += n_al
        return out


ARENA = _ZeroArena()
GRAD_ARENA = _GradArena()
_CONST = {}


def const_tensor(key, values, device, dtype=torch.float32):
    """Small constant tensors (loss weights, safe boxes, channel->head maps) uploaded once instead of one host->device copy per step."""
    k = (key, str(device), dtype)
    t = _CONST.get(k)
    if t is None:
        t = _CONST[k] = torch.tensor(values, dtype=dtype, device=device)
    return t


_BN_TOUCHED = []          # BatchNorm modules that ran in train mode this step (num_batches_tracked bumped once, together)


def begin_step(device):
    # The arenas are recycled here: whatever the weight-gradient stream still has in flight (a backward pass whose end-of-pass join never
    # ran -- an exception inside backward, a pass abandoned by its caller -- or callers that never pass through a join) must be done
    # before the old accumulators are freed and their memory is zero-filled again.  One event record + wait per step.
    if device.type == "cuda" and _WGRAD_STREAMS:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        for d, side in _WGRAD_STREAMS.items():
            if d.index == idx:
                torch.cuda.current_stream(d).wait_stream(side)
        if _PG_KEEP and not _WGRAD_JOIN_QUEUED[0]:
            _PG_KEEP.clear()
    ARENA.begin_step(device)
    GRAD_ARENA.begin_step(device)
    if device.type == "cuda":
        _OPERANDS.refresh_all(device)
    _BN_TOUCHED.clear()
    _PARAM_USES.clear()
    _WGRAD_JOIN_QUEUED[0] = False
    _DEFER_QUEUED[0] = False
    _DEFERRED_LAYOUT.clear()


def end_forward():
    """num_batches_tracked += 1 for every train-mode BatchNorm of this forward in one multi-tensor launch."""
    if _BN_TOUCHED:
        torch._foreach_add_([m.num_batches_tracked for m in _BN_TOUCHED], 1)
        _BN_TOUCHED.clear()


# HIP-graph capture of the dense section (radardistill_amd/graphs.py).  RULE: a captured launch may only touch memory that the
# capture owns (allocated inside it, from the graph's private pool) or that is persistent for the life of the graph (parameters,
# buffers, static inputs, the operand cache's buffers, constants created before the capture).  While CAPTURING is set, zero-filled
# scratch therefore comes from torch.zeros (graph pool + a fill node that re-runs at every replay) instead of the per-step arenas
# -- which are allocated and zeroed by begin_step() OUTSIDE any capture and re-zeroed only up to the eager high-water mark -- and
# the version-keyed weight-layout cache is bypassed (its entries live in the eager pool and are replaced when weights change).
# A plain module-level flag, not thread-local: the captured backward runs on the autograd engine's thread.
CAPTURING = [False]


def zeros_stats(n, device):
    if CAPTURING[0]:
        return torch.zeros(n, dtype=torch.float32, device=device)
    return ARENA.take(n, device)


def zeros_accum(n, device):
    """Zero-filled accumulator whose result may be returned to autograd as a gradient."""
    if CAPTURING[0]:
        return torch.zeros(n, dtype=torch.float32, device=device)
    return GRAD_ARENA.take(n, device)


def kernel_weight(param, Cout, Cin, taps, kind, flip=False):
    """Parameter -> kernel layout [Cout][taps][Cin], cached per parameter version.  Frozen parameters (the teacher) only change
    through torch (load_state_dict bumps `_version`); trainable ones also through the fused optimizer's raw pointers (epoch)."""
    src = param.detach()
    if not src.is_contiguous():
        src = src.contiguous()
    if kind == 0 and not flip:
        return src.reshape(Cout, taps, Cin)       # spconv layout [Cout,kh,kw,Cin] and nn.Linear [Cout,Cin] are already kernel layout
    if not param.is_leaf or CAPTURING[0]:         # a per-step tensor (e.g. the concatenated head-branch weights): nothing to cache on;
        return K.weight_layout(src, Cout, Cin, taps, kind, flip)      # under capture: converted by a node of the graph, every replay
    key = (id(param), kind, flip)
    ver = (param._version, _WEIGHTS_EPOCH[0] if param.requires_grad else _FROZEN_EPOCH[0], param.data_ptr())
    hit = _LAYOUT_CACHE.get(key)
    if hit is not None and hit[0] == ver and hit[2]() is param:      # the weakref guards against id() reuse after a model is freed
        return hit[1]
    w = K.weight_layout(src, Cout, Cin, taps, kind, flip)
    _LAYOUT_CACHE[key] = (ver, w, weakref.ref(param))
    return w


# bf16x3 mode, optional (RD_PRESPLIT=1): operands are split into bf16 hi/lo ONCE per tensor (kernels.split_bf16 /
# weight_layout_split) instead of inside every K step of every GEMM that reads them.  Measured (round 1): the GEMMs gain 3-11 %
# (1.0 ms/step) but the stand-alone split passes cost 2.0 ms/step (139 + 108 launches), so it stays off until the split is fused
# into the producing kernels' epilogues; results are bit-identical either way (tests/test_gpu_model.py).
PRESPLIT = os.environ.get("RD_PRESPLIT", "0") == "1"
_SPLIT_W_CACHE = {}


def kernel_weight_split(param, wk, Cout, Cin, taps, kind2=False):
    """Split-format copy of the kernel-layout weights `wk` of `param` (kind2: the [Cin][taps][Cout] data-gradient layout), cached per
    parameter version like kernel_weight."""
    def make():
        return K.weight_layout_split(wk.contiguous(), Cout, Cin, taps, 2 if kind2 else 0, False)
    if not param.is_leaf:
        return make()
    key = (id(param), kind2)
    ver = (param._version, _WEIGHTS_EPOCH[0] if param.requires_grad else _FROZEN_EPOCH[0], param.data_ptr(), Cout, Cin, taps)
    hit = _SPLIT_W_CACHE.get(key)
    if hit is not None and hit[0] == ver and hit[2]() is param:
        return hit[1]
    w = make()
    _SPLIT_W_CACHE[key] = (ver, w, weakref.ref(param))
    return w


# bf16x3 mode, default: only the WEIGHTS are pre-split.  Trainable weights are re-laid-out once per step anyway (torch layout ->
# [Cout][taps][Cin] for the forward, -> [Cin][taps][Cout] for the data gradient); that launch now writes split format straight from the
# parameter, so the MFMA kernels skip the weight split in every K step at no extra pass (PMC: the split + addressing VALU work was
# 6.7 instructions per MFMA in the dense 3x3 kernel).  Frozen (teacher) weights are converted once.
WSPLIT = os.environ.get("RD_WSPLIT", "1") != "0"
_DGRAD_KIND = {0: 2, 1: 7, 3: 8}


def _b3_wsplit(Cin, Cout):
    """Cin / Cout of the GEMM being launched (the data gradient swaps them)."""
    return WSPLIT and not PRESPLIT and K.get_conv_math() == "bf16x3" and Cout > 32 and Cin % 4 == 0


class _OperandCache:
    """Split-format GEMM operands of leaf parameters in PERSISTENT buffers.  Entries that went stale (the fused optimizer bumped the
    weights epoch, or torch bumped param._version) are re-converted together by refresh_all() -- one launch per step, issued from
    begin_step() -- instead of one small launch per weight per direction (~106 per step); a request for a stale or new entry outside
    that rhythm converts it alone.  Refreshing in place is safe: every reader of the previous contents was enqueued earlier on the
    same (main) stream; the teacher's stream only reads frozen entries, which are never rewritten."""

    def __init__(self):
        self.entries = {}        # (id(param), kind) -> [ver, dst, weakref(param), Cout, Cin, taps, kind]
        self.table = None        # (signature, jobs_dev, chunk_job_dev, chunk_group_dev, n_chunks, keys)

    @staticmethod
    def _ver(param):
        return (param._version, _WEIGHTS_EPOCH[0] if param.requires_grad else _FROZEN_EPOCH[0], param.data_ptr())

    def get(self, param, Cout, Cin, taps, kind):
        key = (id(param), kind)
        e = self.entries.get(key)
        ver = self._ver(param)
        if e is not None and e[2]() is param and (e[3], e[4], e[5]) == (Cout, Cin, taps):
            if e[0] != ver:
                src = param.detach()
                K.weight_layout_split(src if src.is_contiguous() else src.contiguous(), Cout, Cin, taps, kind, False, out=e[1])
                e[0] = ver
            return e[1]
        src = param.detach()
        dst = K.weight_layout_split(src if src.is_contiguous() else src.contiguous(), Cout, Cin, taps, kind, False)
        self.entries[key] = [ver, dst, weakref.ref(param), Cout, Cin, taps, kind]
        self.table = None
        return dst

    def refresh_all(self, device):
        """Re-convert every stale entry of a live, contiguous parameter on `device` in one launch."""
        stale = []
        for key, e in list(self.entries.items()):
            p = e[2]()
            if p is None:
                del self.entries[key]
                self.table = None
                continue
            if p.device == device and p.is_contiguous() and e[0] != self._ver(p):
                stale.append((key, e, p))
        if not stale:
            return
        sig = tuple((key, p.data_ptr()) for key, e, p in stale)
        if self.table is None or self.table[0] != sig:
            import numpy as np
            from .native import LayoutJob
            jobs = (LayoutJob * len(stale))()
            cj, cg = [], []
            for i, (key, e, p) in enumerate(stale):
                jobs[i].src, jobs[i].dst = p.data_ptr(), e[1].data_ptr()
                jobs[i].Cout, jobs[i].Cin, jobs[i].taps, jobs[i].kind = e[3], e[4], e[5], e[6]
                items = np.arange(K.native.lib().rd_weight_layout_split_items(e[3], e[4], e[5], e[6]), dtype=np.int32)
                cj.append(np.full(items.shape, i, dtype=np.int32))
                cg.append(items)
            raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(device)
            cj_d = torch.from_numpy(np.concatenate(cj)).to(device)
            cg_d = torch.from_numpy(np.concatenate(cg)).to(device)
            self.table = (sig, raw, cj_d, cg_d, int(cj_d.numel()))
        _, raw, cj_d, cg_d, n = self.table
        K.weight_layout_split_multi(raw, cj_d, cg_d, n)
        for key, e, p in stale:
            e[0] = self._ver(p)


_OPERANDS = _OperandCache()


def operand_weight_split(param, Cout, Cin, taps, param_kind, dgrad=False, frag=False):
    """Parameter (layout `param_kind`: 0 [Cout][taps][Cin], 1 torch Conv2d, 3 torch ConvTranspose2d) -> split-format GEMM operand:
    [Cout][taps][Cin] for the forward, [Cin][taps][Cout] for the data gradient.  Leaf parameters go through the persistent operand
    cache (one refresh launch per step for all of them); a per-step tensor (the concatenated CenterHead branches) is converted here.
    frag: fragment-major split format (kernels.wants_frag_weights says when the launch takes it)."""
    kind = (_DGRAD_KIND[param_kind] if dgrad else param_kind) | (K.LAYOUT_FRAG if frag else 0)
    if not param.is_leaf:
        src = param.detach()
        return K.weight_layout_split(src if src.is_contiguous() else src.contiguous(), Cout, Cin, taps, kind, False)
    return _OPERANDS.get(param, Cout, Cin, taps, kind)


def split_activation(x):
    """Split-format copy of an activation / gradient tensor, remembered on the tensor object while it is unchanged."""
    hit = getattr(x, "_rd_split", None)
    if hit is not None and hit[0] == x._version and hit[1].data_ptr() != 0:
        return hit[1]
    xs = K.split_bf16(x)
    try:
        x._rd_split = (x._version, xs)
    except Exception:
        pass
    return xs


def _b3_presplit(Cin, Cout, mode):
    return PRESPLIT and K.get_conv_math() == "bf16x3" and Cout > 32 and Cin % 4 == 0 and mode != 3


# Weight gradients on a side HIP stream: dgrad feeds the next (earlier) layer's backward, wgrad feeds nobody until the optimizer, so
# the two GEMMs of a layer need not serialise.  wgrad (+ its layout transform) is launched on a second stream after that stream
# waited for the producer of grad_out; a callback queued on the autograd engine joins the streams when the backward pass ends, so
# every consumer of .grad on the main stream is safe.  Off under DistributedDataParallel (its hooks read gradients mid-backward).
WGRAD_STREAM = [os.environ.get("RD_WGRAD_STREAM", "1") != "0"]
_WGRAD_STREAMS = {}
_WGRAD_JOIN_QUEUED = [False]


def _wgrad_stream(device):
    if not WGRAD_STREAM[0] or device.type != "cuda":
        return None
    s = _WGRAD_STREAMS.get(device)
    if s is None:
        s = _WGRAD_STREAMS[device] = torch.cuda.Stream(device, priority=int(os.environ.get("RD_WGRAD_PRIO", "0")))
    return s


def _queue_wgrad_join(device, main):
    if _WGRAD_JOIN_QUEUED[0]:
        return
    _WGRAD_JOIN_QUEUED[0] = True

    def _join():
        _WGRAD_JOIN_QUEUED[0] = False
        main.wait_stream(_WGRAD_STREAMS[device])
        _PG_KEEP.clear()          # the main stream is now ordered after everything the side stream read

    torch.autograd.Variable._execution_engine.queue_callback(_join)


_PG_STATE = {}          # device -> (main stream of this backward pass, its raw handle, the side stream's raw handle)
# Tensors the side stream reads are HELD until the join instead of being marked with Tensor.record_stream: a recorded block costs the
# caching allocator an event record when it is freed and an event query per later allocation (~200 tensors per step: 0.5 ms of HIP
# runtime calls); with 288 GB of HBM the few GB of gradients that stay alive until the end of the backward pass are free.
_PG_KEEP = []
_set_raw_stream = getattr(torch._C, "_cuda_setStream", None)


def _set_stream(s):
    if _set_raw_stream is not None:
        _set_raw_stream(stream_id=s.stream_id, device_index=s.device_index, device_type=s.device_type)
    else:
        torch.cuda.set_stream(s)


# How often each parameter was consumed by a Function of THIS step's graph (id -> count; reset by begin_step, counted in the
# forwards below).  A parameter used more than once receives the SUM of its gradients: the autograd engine adds them on the main
# stream the moment the second one exists -- long before the join -- so neither may come from the side stream or be a deferred
# (still empty) tensor.
_PARAM_USES = {}


def note_param_use(*params):
    for p in params:
        if p is not None and p.requires_grad:
            _PARAM_USES[id(p)] = _PARAM_USES.get(id(p), 0) + 1


def _side_ok(param):
    """A gradient produced on the side stream must not be READ on the main stream before the join at the end of the backward pass.
    Autograd only aliases it (AccumulateGrad steals a fresh, layout-conforming tensor; CatBackward hands out views) when the
    receiving leaf has no .grad yet; with gradient accumulation it runs `p.grad += g` on the main stream, so those cases stay on
    the main stream.  A non-leaf weight (the concatenated CenterHead branches) names its leaves in `_rd_leaves`."""
    if param is None:
        return True
    if _PARAM_USES.get(id(param), 0) > 1 or getattr(param, "_backward_hooks", None):
        return False          # (a tensor hook reads the gradient on the main stream the moment backward returns it)
    leaves = getattr(param, "_rd_leaves", None)
    if leaves is not None:
        return (not param.is_leaf or param.grad is None) and all(p.grad is None for p in leaves)
    return param.is_leaf and param.grad is None


# A parameter gradient that reaches its leaves without autograd's AccumulateGrad nodes (ConcatLeaves below) still has to reach whoever
# listens for "this parameter's gradient is complete" (FlatAdam's bucketed all-reduce): listeners are called with the list of leaves.
GRAD_LISTENERS = []


def deliver_grads(leaves, grads):
    """leaf.grad = g (or += g when the leaf already holds a gradient) for every pair, then tell the listeners once with the list."""
    for leaf, g in zip(leaves, grads):
        if leaf.grad is None:
            leaf.grad = g
        else:
            leaf.grad = leaf.grad + g
    for cb in GRAD_LISTENERS:
        cb(leaves)


# Off under torch's DistributedDataParallel (dist.data_parallel, RD_DDP=torch): its reducer only sees gradients that pass through the
# parameters' own AccumulateGrad nodes; gradients handed over as views of a concatenated leaf would count as "unused parameters".
CONCAT_LEAVES = [True]


class ConcatLeaves:
    """Several leaf parameters that one kernel consumes concatenated along dim 0 (the 42 CenterHead branches: weights, biases,
    BatchNorm affines), kept as ONE persistent leaf tensor: refresh() copies the current parameter values in (one launch), the
    convolution / BatchNorm Functions see a single leaf (one AccumulateGrad node, the persistent operand cache applies), and when its
    gradient has been accumulated a hook hands dim-0 VIEWS of it to the real parameters.  Before: torch.cat in the graph, i.e. a
    CatBackward (42 narrows) + 42 AccumulateGrad nodes per concatenated tensor and step -- 1.2 ms of host time per step for the head."""

    def __init__(self, leaves):
        self.leaves = list(leaves)
        self.sizes = [int(p.shape[0]) for p in self.leaves]
        with torch.no_grad():
            self.cat = torch.cat([p.detach() for p in self.leaves], 0).contiguous()
        self.cat.requires_grad_(True)
        self.cat._rd_leaves = self.leaves
        self.cat.register_post_accumulate_grad_hook(self._distribute)
        self._sig = None

    def valid(self):
        c = self.cat
        return all(p.requires_grad and p.device == c.device and p.dtype == c.dtype for p in self.leaves)

    def refresh(self):
        """Copy the parameters' current values into the concatenated leaf when any of them changed -> the leaf."""
        self.cat.grad = None          # last pass's gradient lives on in the leaves' views
        sig = (tuple(p._version for p in self.leaves), _WEIGHTS_EPOCH[0], self.leaves[0].data_ptr())
        if sig != self._sig:
            with torch.no_grad():
                torch.cat([p.detach() for p in self.leaves], 0, out=self.cat)
            self._sig = sig
        return self.cat

    def _distribute(self, cat):
        # cat.grad itself stays until the next refresh(): a deferred weight re-layout finds its destination through it
        deliver_grads(self.leaves, cat.grad.split(self.sizes))


def param_grad_stream(fn, *inputs, param=None):
    """Run `fn()` -- a launch sequence that only produces PARAMETER gradients (nothing later in this backward pass reads its result)
    -- on the weight-gradient side stream; `inputs` are the tensors it reads (kept from being recycled under it).  Falls back to a
    plain call when the side stream is off or `param` (the tensor receiving the gradient) may be read on the main stream (_side_ok).
    Must be called from inside an autograd Function's backward.
    Called ~150 times per step, so it avoids the Python-heavy torch.cuda helpers (current_stream / wait_stream / the stream context
    manager were ~40 us per call, 6 ms of host time per step): the main stream is looked up once per backward pass, the fork is one
    C call (rd_stream_fork: event record + stream wait), and the current stream is switched through the raw setter."""
    dev = inputs[0].device
    side = _wgrad_stream(dev)
    if side is None or not _side_ok(param):
        return fn()
    st = _PG_STATE.get(dev)
    if st is None or not _WGRAD_JOIN_QUEUED[0]:
        main = torch.cuda.current_stream(dev)
        st = _PG_STATE[dev] = (main, main.cuda_stream, side.cuda_stream)
        _queue_wgrad_join(dev, main)
    main, main_raw, side_raw = st
    K.check(K.native.lib().rd_stream_fork(main_raw, side_raw), "rd_stream_fork")          # the side stream waits for the main stream's work so far
    _set_stream(side)
    try:
        out = fn()
        if torch.is_tensor(out) and not out.is_contiguous():
            out = out.contiguous()            # a strided gradient would be CLONED by AccumulateGrad -- on the main stream, unordered
    finally:
        _set_stream(main)
    _PG_KEEP.append(inputs)
    return out


# Weight gradients leave the GEMM kernels in kernel layout [Cout][taps][Cin]; nn.Conv2d / nn.ConvTranspose2d parameters want
# [Cout][Cin][kh][kw] / [Cin][Cout][kh][kw].  Instead of one small re-layout launch per layer (49 per step), the destination tensor is
# handed to autograd EMPTY and all of a backward pass's re-layouts run as ONE launch when the pass ends (engine callback), on the
# weight-gradient stream before it joins the main stream.  Same precondition as the side stream itself (_side_ok): nothing reads the
# gradient before the end of the pass -- AccumulateGrad only stores a fresh gradient when the leaf has no .grad yet.
_DEFERRED_LAYOUT = []
_DEFER_QUEUED = [False]
DEFER_LAYOUT = [os.environ.get("RD_DEFER_LAYOUT", "1") != "0"]


def _defer_layout_ok(param):
    """Deferral hands autograd a gradient tensor that is still EMPTY until the pass ends, so nothing may read or combine it before
    then: not a second use of the same weight in this graph (the engine would sum two empty tensors and both jobs would fill one
    destination -- the later occurrences are laid out immediately instead), not a tensor hook on the parameter (it would see, scale
    or clip garbage)."""
    if not (DEFER_LAYOUT[0] and WGRAD_STREAM[0] and param.is_cuda and _side_ok(param)):
        return False
    if getattr(param, "_backward_hooks", None):
        return False
    return all(j[2] is not param for j in _DEFERRED_LAYOUT)


def _flush_deferred_layouts(only_accumulated=False):
    """only_accumulated: mid-backward flush (a gradient bucket is about to be packed): only the jobs whose parameter already HAS its
    gradient -- for the others AccumulateGrad has not run yet and may still decide to clone the destination."""
    from .native import LayoutJob
    if only_accumulated:
        jobs = [j for j in _DEFERRED_LAYOUT if j[2].grad is not None]
        _DEFERRED_LAYOUT[:] = [j for j in _DEFERRED_LAYOUT if j[2].grad is None]
    else:
        jobs, _DEFERRED_LAYOUT[:] = list(_DEFERRED_LAYOUT), []
    for i in range(0, len(jobs), 96):
        part = jobs[i:i + 96]
        arr = (LayoutJob * len(part))()
        n = 0
        for (src, dst_ref, param, Cout, Cin, taps, kind) in part:
            # where the gradient ended up: AccumulateGrad normally keeps the very tensor backward returned (then .grad IS the deferred
            # destination); had it cloned instead (an extra reference, a layout mismatch), the clone is what must be filled.  Without a
            # .grad the destination is only written while it is still alive (weak reference) -- never through a stale address.
            g = param.grad
            if g is None:
                g = dst_ref()
                if g is None:
                    continue
            arr[n].src, arr[n].dst = src.data_ptr(), g.data_ptr()
            arr[n].Cout, arr[n].Cin, arr[n].taps, arr[n].kind = Cout, Cin, taps, kind
            n += 1
        if n:
            K.weight_layout_multi(arr, n)


def defer_weight_layout(gwk, param, Cout, Cin, taps, kind):
    """-> the (still unwritten) gradient tensor in the parameter's layout; filled by _flush_deferred_layouts at the end of the pass.
    Called on the weight-gradient stream (inside param_grad_stream) or, without it, on the main stream.  No reference to the
    returned tensor is kept here: AccumulateGrad only adopts a gradient nobody else holds (it clones otherwise)."""
    dst = torch.empty(tuple(param.shape), dtype=torch.float32, device=gwk.device)
    _DEFERRED_LAYOUT.append((gwk, weakref.ref(dst), param, Cout, Cin, taps, kind))
    if not _DEFER_QUEUED[0]:
        _DEFER_QUEUED[0] = True
        dev = gwk.device

        def _finish():
            _DEFER_QUEUED[0] = False
            side = _WGRAD_STREAMS.get(dev) if WGRAD_STREAM[0] else None
            if side is not None:
                cur = torch.cuda.current_stream(dev)
                _set_stream(side)
                try:
                    _flush_deferred_layouts()
                finally:
                    _set_stream(cur)
                cur.wait_stream(side)
            else:
                _flush_deferred_layouts()

        torch.autograd.Variable._execution_engine.queue_callback(_finish)
    return dst


class ConvSpec:
    """Geometry of one convolution call: how output rows find their input rows, forward and backward.

    fwd_ix / bwd_ix are rd_conv_index structs (kernels.conv_index_*); keep holds tensors referenced by raw pointer.
    param_kind: 0 spconv / linear layout, 1 torch Conv2d layout, 3 torch ConvTranspose2d layout.
    """

    def __init__(self, taps, in_rows, out_rows, fwd_ix, bwd_ix, param_kind, keep=(), fwd_nbr=None, bwd_nbr=None):
        self.taps, self.in_rows, self.out_rows = taps, in_rows, out_rows
        self.fwd_ix, self.bwd_ix, self.param_kind = fwd_ix, bwd_ix, param_kind
        self.keep = keep
        self.fwd_nbr, self.bwd_nbr = fwd_nbr, bwd_nbr


# Small per-layer host costs (round 3, measured together by step-by-step alternation, tools/diag/toggle_ab.py): the geometry spec of a
# dense layer and of a linear layer is built once per shape instead of once per call (49 + 21 calls per step), and a BatchNorm module's
# four tensors are looked up in its own dictionaries instead of through nn.Module.__getattr__ (4 slow attribute reads per layer).
HOT_CACHES = [os.environ.get("RD_HOT_CACHES", "1") != "0"]
_DENSE_SPECS, _LINEAR_SPECS = {}, {}


def dense_conv_spec(B, Hin, Win, KH, KW, stride, pad, transposed=False):
    if HOT_CACHES[0]:
        key = (B, Hin, Win, KH, KW, stride, pad, transposed)
        spec = _DENSE_SPECS.get(key)
        if spec is None:
            if len(_DENSE_SPECS) > 512:
                _DENSE_SPECS.clear()
            spec = _DENSE_SPECS[key] = _dense_conv_spec(B, Hin, Win, KH, KW, stride, pad, transposed)
        return spec
    return _dense_conv_spec(B, Hin, Win, KH, KW, stride, pad, transposed)


def _dense_conv_spec(B, Hin, Win, KH, KW, stride, pad, transposed=False):
    if not transposed:
        Hout = (Hin + 2 * pad - KH) // stride + 1
        Wout = (Win + 2 * pad - KW) // stride + 1
        fwd = K.conv_index_dense(B, Hin, Win, Hout, Wout, KH, KW, stride, pad, transposed=False)
        bwd = K.conv_index_dense(B, Hout, Wout, Hin, Win, KH, KW, stride, pad, transposed=True)
        kind = 1
    else:
        Hout = (Hin - 1) * stride - 2 * pad + KH
        Wout = (Win - 1) * stride - 2 * pad + KW
        fwd = K.conv_index_dense(B, Hin, Win, Hout, Wout, KH, KW, stride, pad, transposed=True)
        bwd = K.conv_index_dense(B, Hout, Wout, Hin, Win, KH, KW, stride, pad, transposed=False)
        kind = 3
    spec = ConvSpec(KH * KW, B * Hin * Win, B * Hout * Wout, fwd, bwd, kind)
    spec.out_hw = (Hout, Wout)
    return spec


def linear_spec(rows):
    if HOT_CACHES[0]:
        spec = _LINEAR_SPECS.get(rows)
        if spec is None:
            if len(_LINEAR_SPECS) > 512:
                _LINEAR_SPECS.clear()
            spec = _LINEAR_SPECS[rows] = _linear_spec(rows)
        return spec
    return _linear_spec(rows)


def _linear_spec(rows):
    fwd = K.conv_index_dense(1, rows, 1, rows, 1, 1, 1, 1, 0)
    bwd = K.conv_index_dense(1, rows, 1, rows, 1, 1, 1, 1, 0)
    return ConvSpec(1, rows, rows, fwd, bwd, 0)


class _ConvFn(torch.autograd.Function):
    """out = conv(x, W) + b with optional fused epilogue (eval-BN scale/shift, residual, ReLU) when no grad is needed,
    or fused BatchNorm statistics (stats) in training."""

    @staticmethod
    def forward(ctx, x, weight, bias, spec, Cout, stats, bias_feeds_bn=False, residual=None):
        out = _ConvFn.forward_impl(ctx, x, weight, bias, spec, Cout, stats, residual)
        ctx.bias_feeds_bn = bool(bias_feeds_bn)
        ctx.save_for_backward(x, weight)
        return out

    @staticmethod
    def forward_impl(ctx, x, weight, bias, spec, Cout, stats, residual=None):
        """The convolution launch + what its backward needs on `ctx` (shared with _ConvBNActFn); the caller saves x and weight.
        residual (rows x Cout): added in the kernel's epilogue (out = conv + bias + residual; its gradient is grad_out itself)."""
        Cin = x.shape[1]
        note_param_use(weight, bias)
        ctx.xs = None
        if _b3_wsplit(Cin, Cout):
            wk = None
            frag = K.wants_frag_weights(spec.fwd_ix, spec.in_rows, spec.out_rows, Cin, Cout, spec.taps)
            out = K.conv_fwd(x, operand_weight_split(weight, Cout, Cin, spec.taps, spec.param_kind, frag=frag), spec.taps, bias, spec.out_rows, Cout,
                             spec.fwd_ix, stats=stats, nbr_keepalive=spec.fwd_nbr, w_split=2 if frag else True, residual=residual)
            ctx.spec, ctx.Cout, ctx.Cin = spec, Cout, Cin
            ctx.has_bias = bias is not None
            ctx.bias_ref = bias
            ctx.wk = None
            return out
        wk = kernel_weight(weight, Cout, Cin, spec.taps, spec.param_kind)
        if _b3_presplit(Cin, Cout, spec.fwd_ix.mode):
            ctx.xs = split_activation(x)              # reused by the weight gradient
            out = K.conv_fwd(ctx.xs, kernel_weight_split(weight, wk, Cout, Cin, spec.taps), spec.taps, bias, spec.out_rows, Cout, spec.fwd_ix,
                             stats=stats, nbr_keepalive=spec.fwd_nbr, in_split=True, w_split=True, residual=residual)
        else:
            out = K.conv_fwd(x, wk, spec.taps, bias, spec.out_rows, Cout, spec.fwd_ix, stats=stats, nbr_keepalive=spec.fwd_nbr, residual=residual)
        ctx.spec, ctx.Cout, ctx.Cin = spec, Cout, Cin
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        ctx.wk = wk                                # kernel-layout weights of THIS step (the optimizer runs after backward)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        go = grad_out.contiguous()
        gx, gw, gb = _ConvFn.backward_impl(ctx, x, weight, go, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        n = ctx.needs_input_grad
        if len(n) == 7:          # called without a residual
            return gx, gw, gb, None, None, None, None
        return gx, gw, gb, None, None, None, None, (go if n[7] else None)

    @staticmethod
    def backward_impl(ctx, x, weight, grad_out, need_x, need_w, need_b):
        spec, Cout, Cin = ctx.spec, ctx.Cout, ctx.Cin
        gx = gw = gb = None
        if need_x:
            wk = ctx.wk
            use_ws = wk is None and Cout % 32 == 0 and _b3_wsplit(Cout, Cin)
            if wk is None and not use_ws:
                wk = kernel_weight(weight, Cout, Cin, spec.taps, spec.param_kind)
            # exact fp32: the kernel reads the forward weights transposed (no re-layout launch).  bf16x3: the transposed read costs
            # a register transpose per weight tile and measured ~1 ms/step slower than re-laying the weights out once, so that
            # mode keeps the [Cin][taps][Cout] copy (rd_conv_dgrad itself works in both modes).
            if use_ws:
                # bf16x3: [Cin][taps][Cout] split-format operand straight from the parameter (unchanged since the forward: the
                # optimizer runs after backward)
                frag = K.wants_frag_weights(spec.bwd_ix, spec.out_rows, spec.in_rows, Cout, Cin, spec.taps)
                wds = operand_weight_split(weight, Cout, Cin, spec.taps, spec.param_kind, dgrad=True, frag=frag)
                gx = K.conv_fwd(grad_out, wds, spec.taps, None, spec.in_rows, Cin, spec.bwd_ix, nbr_keepalive=spec.bwd_nbr, w_split=2 if frag else True)
            elif Cout % 32 == 0 and K.get_conv_math() == "f32":
                gx = K.conv_dgrad(grad_out, wk, spec.taps, spec.in_rows, Cin, spec.bwd_ix, nbr_keepalive=spec.bwd_nbr)   # forward weights, read transposed
            elif Cout % 32 == 0 and _b3_presplit(Cout, Cin, spec.bwd_ix.mode):
                # bf16x3 with pre-split operands: grad_out is split once (shared with the weight gradient below), the weights are
                # re-laid-out to [Cin][taps][Cout] and split in one launch
                gos = split_activation(grad_out)
                wds = kernel_weight_split(weight, wk, Cout, Cin, spec.taps, kind2=True)
                gx = K.conv_fwd(gos, wds, spec.taps, None, spec.in_rows, Cin, spec.bwd_ix, nbr_keepalive=spec.bwd_nbr, in_split=True, w_split=True)
            else:
                go, Cp = grad_out, Cout
                if Cout % 32 != 0:      # narrow outputs (27-channel DCN offsets): zero-pad the contraction dim to the kernel's K step
                    Cp = (Cout + 31) // 32 * 32
                    go = torch.nn.functional.pad(grad_out, (0, Cp - Cout))
                    wk = torch.nn.functional.pad(wk.reshape(Cout, -1), (0, 0, 0, Cp - Cout))
                wd = K.weight_layout(wk.contiguous(), Cp, Cin, spec.taps, 2, False)          # [Cin][taps][Cout]
                gx = K.conv_fwd(go, wd, spec.taps, None, spec.in_rows, Cin, spec.bwd_ix, nbr_keepalive=spec.bwd_nbr)
            if _DEBUG and spec.fwd_nbr is not None:
                w3 = kernel_weight(weight, Cout, Cin, spec.taps, spec.param_kind).reshape(-1, spec.taps, Cin)[:Cout].double()
                ref = torch.zeros((spec.in_rows, Cin), dtype=torch.float64, device=x.device)
                nb = spec.fwd_nbr.long()
                for t in range(spec.taps):
                    o = torch.nonzero(nb[:, t] >= 0).squeeze(1)
                    ref.index_add_(0, nb[o, t], grad_out[o].double() @ w3[:, t, :])
                _dbg_report(f"conv dgrad Cin={Cin} Cout={Cout} rows {spec.out_rows}->{spec.in_rows} flip={spec.bwd_ix.flip}", gx.double(), ref)
        if need_w:
            gw = param_grad_stream(lambda: _ConvFn._wgrad(ctx, x, weight, grad_out, spec, Cout, Cin), x, grad_out, param=weight)
        if ctx.has_bias and need_b:
            if getattr(ctx, 'bias_feeds_bn', False):
                # The conv output goes only into a train-mode BatchNorm, which subtracts the batch mean: d loss / d bias = sum over rows
                # of the BatchNorm input gradient = gamma * rstd * (sum g - sum g - sum(xhat) * dgamma / n) = 0 identically (sum(xhat)
                # = 0).  The reference evaluates that sum numerically and gets rounding noise (~1e-7 of the weight gradients); here
                # it is the exact value, at no launch (was: one column-sum kernel + one stream fork per layer, 45 per step).
                gb = zeros_accum(Cout, grad_out.device)
            else:
                gb = param_grad_stream(lambda: K.colsum(grad_out) if Cout % 4 == 0 else grad_out.sum(0), grad_out, param=ctx.bias_ref)
        return gx, gw, gb

    @staticmethod
    def _wgrad(ctx, x, weight, grad_out, spec, Cout, Cin):
        if True:
            if PRESPLIT and K.get_conv_math() == "bf16x3" and Cout >= 64 and Cin >= 64 and Cout % 4 == 0:
                xs = ctx.xs if ctx.xs is not None else (split_activation(x) if spec.fwd_ix.mode != 3 else None)
                gwk = K.conv_wgrad(xs if xs is not None else x, split_activation(grad_out), spec.taps, spec.fwd_ix, nbr_keepalive=spec.fwd_nbr,
                                   in_split=xs is not None, go_split=True)
            else:
                gwk = K.conv_wgrad(x, grad_out, spec.taps, spec.fwd_ix, nbr_keepalive=spec.fwd_nbr)         # kernel layout
            if spec.param_kind == 0:
                gw = gwk.reshape(weight.shape)
            elif weight.is_leaf and _defer_layout_ok(weight):
                gw = defer_weight_layout(gwk, weight, Cout, Cin, spec.taps, 4 if spec.param_kind == 1 else 5)
            elif spec.param_kind == 1:
                gw = K.weight_layout(gwk, Cout, Cin, spec.taps, 4, False, out_shape=tuple(weight.shape))
            else:
                gw = K.weight_layout(gwk, Cout, Cin, spec.taps, 5, False, out_shape=tuple(weight.shape))
            return gw


def conv(x, weight, bias, spec, Cout, stats=None, bias_feeds_bn=False, residual=None):
    """bias_feeds_bn: the output is consumed only by a train-mode BatchNorm (whose statistics `stats` collects): the bias gradient is
    then identically zero and no column sum is launched for it.  residual: added in the epilogue (one launch and one autograd node
    less than `conv(...) + residual`)."""
    if residual is None:
        return _ConvFn.apply(x, weight, bias, spec, Cout, stats, bias_feeds_bn)
    return _ConvFn.apply(x, weight, bias, spec, Cout, stats, bias_feeds_bn, residual)


def conv_inference(x, weight, bias, spec, Cout, scale=None, shift=None, residual=None, relu=False):
    """Frozen path (teacher): conv + folded eval-mode BatchNorm + residual + ReLU in ONE kernel, no graph."""
    Cin = x.shape[1]
    if _b3_wsplit(Cin, Cout):
        frag = K.wants_frag_weights(spec.fwd_ix, spec.in_rows, spec.out_rows, Cin, Cout, spec.taps)
        return K.conv_fwd(x, operand_weight_split(weight, Cout, Cin, spec.taps, spec.param_kind, frag=frag), spec.taps, bias, spec.out_rows, Cout,
                          spec.fwd_ix, scale=scale, shift=shift, residual=residual, relu=relu, nbr_keepalive=spec.fwd_nbr,
                          w_split=2 if frag else True)
    wk = kernel_weight(weight, Cout, Cin, spec.taps, spec.param_kind)
    if _b3_presplit(Cin, Cout, spec.fwd_ix.mode):
        return K.conv_fwd(split_activation(x), kernel_weight_split(weight, wk, Cout, Cin, spec.taps), spec.taps, bias, spec.out_rows, Cout,
                          spec.fwd_ix, scale=scale, shift=shift, residual=residual, relu=relu, nbr_keepalive=spec.fwd_nbr,
                          in_split=True, w_split=True)
    return K.conv_fwd(x, wk, spec.taps, bias, spec.out_rows, Cout, spec.fwd_ix, scale=scale, shift=shift, residual=residual,
                      relu=relu, nbr_keepalive=spec.fwd_nbr)


SYNC_BN = [False]          # train.convert_sync_batchnorm(model): every train-mode BatchNorm reduces its batch statistics over the group


def sync_group(bn=None):
    """Process group a train-mode BatchNorm synchronises over, or None: torch.nn.SyncBatchNorm modules (the reference's --sync_bn,
    tools/train.py:144-145) and/or the global switch, and only when a process group with more than one rank exists."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if isinstance(bn, torch.nn.SyncBatchNorm):
        group = bn.process_group if bn.process_group is not None else dist.group.WORLD
    elif SYNC_BN[0]:
        group = dist.group.WORLD
    else:
        return None
    return group if dist.get_world_size(group) > 1 else None


def _group_sum(group):
    import torch.distributed as dist
    return lambda t: dist.all_reduce(t, group=group)


def synced_stats(stats_ext, C, rows, group):
    """stats_ext (2C + 1): local (sum, sumsq) already in [0, 2C) -> row count appended, summed over the group in place."""
    stats_ext[2 * C:].fill_(float(rows))
    _group_sum(group)(stats_ext)
    return stats_ext


class _BNActFn(torch.autograd.Function):
    """y = act(batchnorm_train(x) [+ residual]) over rows; batch statistics either precomputed by the producing conv
    kernel's epilogue (`stats`) or computed here.  group: SyncBatchNorm process group (stats then has room for 2C + 1 values)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, eps, momentum, act, stats, group=None):
        rows, C = x.shape
        if group is not None:
            if stats is None:
                stats = K.bn_stats(x, extra=1)
            stats = synced_stats(stats, C, rows, group)
        elif stats is None:
            stats = K.bn_stats(x)
        y, side = K.bn_train_fwd(x, stats, gamma, beta, eps, momentum, running_mean, running_var, residual, act, sync=group is not None)
        ctx.act, ctx.has_res, ctx.group = act, residual is not None, group
        ctx.count = stats[2 * C:] if group is not None else None          # a slice of the per-step arena (shared version counter): kept as an attribute
        ctx.save_for_backward(x, y, gamma, side)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, gamma, side = ctx.saved_tensors
        sync = (_group_sum(ctx.group), ctx.count) if ctx.group is not None else None
        gx, gres, gg, gb = K.bn_bwd(x, y, gy.contiguous(), gamma, side, ctx.act, ctx.has_res, sync=sync)
        if _DEBUG and ctx.act in (0, 1) and sync is None:
            mean, rstd = side[0], side[1]
            g = gy.double() * ((y > 0).double() if ctx.act == 1 else 1.0)
            xh = (x.double() - mean.double()) * rstd.double()
            n = x.shape[0]
            ref = gamma.double() * rstd.double() * (g - g.sum(0) / n - xh * (g * xh).sum(0) / n)
            _dbg_report(f"bn bwd C={x.shape[1]} rows={n} res={ctx.has_res}", gx.double(), ref)
            if ctx.has_res:
                _dbg_report("bn bwd residual grad", gx.double() * 0 + gres.double(), g)
        return gx, gg, gb, gres, None, None, None, None, None, None, None


def bn_act_train(x, bn, residual=None, act=1, stats=None):
    """bn: nn.BatchNorm1d/2d module (train mode semantics: batch stats, running-stat update)."""
    group = sync_group(bn)
    if x.shape[0] <= 1 and group is None:
        raise ValueError("Expected more than 1 value per channel when training")     # torch's own train-mode BN error
    _BN_TOUCHED.append(bn)
    if group is not None and stats is not None and stats.numel() != 2 * x.shape[1] + 1:
        stats = torch.cat([stats, stats.new_zeros(1)])
    return _BNActFn.apply(x, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, float(bn.eps), float(bn.momentum), act, stats, group)


def bn_act_train_tensors(x, gamma, beta, running_mean, running_var, eps, momentum, act=1, stats=None, modules=()):
    """Train-mode BatchNorm over explicit tensors (several BatchNorm modules batched on the channel axis); `modules` get their
    num_batches_tracked bumped with everything else at end_forward()."""
    group = sync_group(modules[0] if modules else None)
    if x.shape[0] <= 1 and group is None:
        raise ValueError("Expected more than 1 value per channel when training")
    _BN_TOUCHED.extend(modules)
    if group is not None and stats is not None and stats.numel() != 2 * x.shape[1] + 1:
        stats = torch.cat([stats, stats.new_zeros(1)])
    return _BNActFn.apply(x, gamma, beta, None, running_mean, running_var, eps, momentum, act, stats, group)


# One library call per layer and direction (composite.hip).  ON by default (RD_COMPOSITE=0 disables).  Run against run it looked
# neutral (a box's host loop drifts between 13 and 19 ms per step within a minute); alternating the switch step by step inside ONE
# process (tools/diag/toggle_ab.py, 150 + 100 steps per arm, device idle at every step start) it saves 0.36-0.48 ms of host time per
# step: median 18.71 -> 18.35 ms at B = 1, 13.58 -> 13.10 ms at B = 8 (minima 17.63 -> 17.24, 12.52 -> 12.09).
COMPOSITE = [os.environ.get("RD_COMPOSITE", "1") != "0"]


def _side_handles(dev, param):
    """(raw handle of the main stream, raw handle of the weight-gradient stream or None) for a composite backward call; queues the
    end-of-backward join like param_grad_stream.  None for the side stream when it is off or `param` may be read on the main stream
    before the join (_side_ok)."""
    side = _wgrad_stream(dev)
    if side is None or not _side_ok(param):
        return K._stream(), None
    st = _PG_STATE.get(dev)
    if st is None or not _WGRAD_JOIN_QUEUED[0]:
        main = torch.cuda.current_stream(dev)
        st = _PG_STATE[dev] = (main, main.cuda_stream, side.cuda_stream)
        _queue_wgrad_join(dev, main)
    return st[1], st[2]


class _ConvBNActFn(torch.autograd.Function):
    """conv (+bias) -> train-mode BatchNorm (statistics from the conv epilogue) -> (+residual) -> activation as ONE autograd node:
    the same four launches as _ConvFn + _BNActFn (forward: conv, BN apply; backward: BN reduce + apply, data gradient, weight
    gradient), but one Function.apply and one backward node per layer instead of two (~20 us of host time per layer and step).
    group: SyncBatchNorm process group (one small all-reduce in the forward, one in the backward).
    Fast path (COMPOSITE): forward and backward are ONE library call each (rd_conv_bn_act_fwd / _bwd, csrc/composite.hip) -- the
    same launches in the same order on the same streams; the per-launch path below remains for SyncBatchNorm, the debug checks, the
    pre-split experiment and weights the fast path has no operand format for."""

    @staticmethod
    def forward(ctx, x, weight, bias, spec, Cout, gamma, beta, residual, running_mean, running_var, eps, momentum, act, group=None):
        sync = group is not None
        Cin = x.shape[1]
        fmt = None
        if COMPOSITE[0] and not sync and not _DEBUG and x.is_cuda and spec.fwd_ix.mode != 3 and K.BN_PROFILE is None:
            if _b3_wsplit(Cin, Cout):
                frag = K.wants_frag_weights(spec.fwd_ix, spec.in_rows, spec.out_rows, Cin, Cout, spec.taps)
                fmt = 2 if frag else 1
                wop = operand_weight_split(weight, Cout, Cin, spec.taps, spec.param_kind, frag=frag)
            elif not _b3_presplit(Cin, Cout, spec.fwd_ix.mode):
                fmt = 0
                wop = kernel_weight(weight, Cout, Cin, spec.taps, spec.param_kind)
        if fmt is not None:
            note_param_use(weight, bias)
            stats = zeros_stats(2 * Cout, x.device)
            raw, y, side = K.conv_bn_act_fwd(x, wop, fmt, spec.taps, bias, spec.fwd_ix, spec.out_rows, Cout, stats, gamma, beta, eps, momentum,
                                             running_mean, running_var, residual, act, nbr_keepalive=spec.fwd_nbr)
            ctx.spec, ctx.Cout, ctx.Cin = spec, Cout, Cin
            ctx.has_bias, ctx.bias_ref, ctx.bias_feeds_bn = bias is not None, bias, True
            ctx.wk = wop if fmt == 0 else None
            ctx.xs = None
            ctx.fast = True
            ctx.act, ctx.has_res, ctx.group, ctx.count = act, residual is not None, None, None
            ctx.save_for_backward(x, weight, raw, y, gamma, side)
            return y
        ctx.fast = False
        ext = zeros_stats(2 * Cout + (1 if sync else 0), x.device)
        stats = ext[:2 * Cout] if sync else ext
        raw = _ConvFn.forward_impl(ctx, x, weight, bias, spec, Cout, stats)
        ctx.bias_feeds_bn = True
        if sync:
            synced_stats(ext, Cout, raw.shape[0], group)
        y, side = K.bn_train_fwd(raw, ext, gamma, beta, eps, momentum, running_mean, running_var, residual, act, sync=sync)
        ctx.act, ctx.has_res, ctx.group = act, residual is not None, group
        ctx.count = ext[2 * Cout:] if sync else None
        ctx.save_for_backward(x, weight, raw, y, gamma, side)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, raw, y, gamma, side = ctx.saved_tensors
        n = ctx.needs_input_grad
        spec, Cout, Cin = ctx.spec, ctx.Cout, ctx.Cin
        fmt = None
        if ctx.fast and COMPOSITE[0] and K.BN_PROFILE is None:
            # the data-gradient operand the per-launch path would pick (_ConvFn.backward_impl)
            if not n[0]:
                fmt, wd = (1 if K.get_conv_math() == "bf16x3" else 0), None
            elif ctx.wk is None and Cout % 32 == 0 and _b3_wsplit(Cout, Cin):
                frag = K.wants_frag_weights(spec.bwd_ix, spec.out_rows, spec.in_rows, Cout, Cin, spec.taps)
                fmt, wd = (2 if frag else 1), operand_weight_split(weight, Cout, Cin, spec.taps, spec.param_kind, dgrad=True, frag=frag)
            elif ctx.wk is not None and Cout % 32 == 0 and K.get_conv_math() == "f32":
                fmt, wd = 0, ctx.wk
        if fmt is not None:
            gyc = gy if gy.is_contiguous() else gy.contiguous()
            main_raw, side_raw = _side_handles(x.device, weight) if n[1] else (K._stream(), None)
            gx, gres, gg, gb_bn, gwk, graw = K.conv_bn_act_bwd(raw, y, gyc, gamma, side, ctx.act, ctx.has_res, wd, fmt, spec.taps, n[0], spec.in_rows,
                                                               Cin, spec.bwd_ix, x, spec.fwd_ix, n[1], main_raw, side_raw,
                                                               fwd_nbr=spec.fwd_nbr, bwd_nbr=spec.bwd_nbr)
            gw = None
            if n[1]:
                if side_raw is not None:
                    _PG_KEEP.append((x, graw))          # read by the side stream: held until the streams join
                if spec.param_kind == 0:
                    gw = gwk.reshape(weight.shape)
                elif weight.is_leaf and _defer_layout_ok(weight):
                    gw = defer_weight_layout(gwk, weight, Cout, Cin, spec.taps, 4 if spec.param_kind == 1 else 5)
                else:          # rare: the re-layout follows the weight gradient on its stream
                    kind = 4 if spec.param_kind == 1 else 5
                    if side_raw is not None:
                        main = _PG_STATE[x.device][0]
                        _set_stream(_WGRAD_STREAMS[x.device])
                        try:
                            gw = K.weight_layout(gwk, Cout, Cin, spec.taps, kind, False, out_shape=tuple(weight.shape))
                        finally:
                            _set_stream(main)
                    else:
                        gw = K.weight_layout(gwk, Cout, Cin, spec.taps, kind, False, out_shape=tuple(weight.shape))
            # the bias in front of a train-mode BatchNorm has the exact gradient zero (_ConvFn.backward_impl)
            gb = zeros_accum(Cout, gyc.device) if (ctx.has_bias and n[2]) else None
            return gx, gw, gb, None, None, gg, gb_bn, gres, None, None, None, None, None, None
        sync = (_group_sum(ctx.group), ctx.count) if ctx.group is not None else None
        if ctx.fast and ctx.wk is None:
            ctx.wk = None
        graw, gres, gg, gb_bn = K.bn_bwd(raw, y, gy.contiguous(), gamma, side, ctx.act, ctx.has_res, sync=sync)
        gx, gw, gb = _ConvFn.backward_impl(ctx, x, weight, graw, n[0], n[1], n[2])
        return gx, gw, gb, None, None, gg, gb_bn, gres, None, None, None, None, None, None


def conv_bn_act_train(x, weight, bias, spec, Cout, bn, residual=None, act=1):
    """Training-mode conv -> BatchNorm module `bn` -> (+residual) -> act, one autograd node (see _ConvBNActFn)."""
    group = sync_group(bn)
    if spec.out_rows <= 1 and group is None:
        raise ValueError("Expected more than 1 value per channel when training")     # torch's own train-mode BN error
    _BN_TOUCHED.append(bn)
    if HOT_CACHES[0]:
        gamma, beta, rm, rv, eps, mom = bn_tensors(bn)
        return _ConvBNActFn.apply(x, weight, bias, spec, Cout, gamma, beta, residual, rm, rv, eps, mom, act, group)
    return _ConvBNActFn.apply(x, weight, bias, spec, Cout, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, float(bn.eps),
                              float(bn.momentum), act, group)


def fast_module_attrs(model, on=True):
    """Mirror every submodule and parameter of `model` into its owner's instance dictionary, so that `self.conv1` / `conv.weight` are
    plain attribute reads instead of calls of nn.Module.__getattr__ (a Python function searching three dictionaries; ~1200 such
    reads per training step).  nn.Module.__setattr__ removes the mirror entry of a name it re-assigns, Module.to() and
    load_state_dict() keep parameter objects, buffers are NOT mirrored (Module.to() replaces them); register_parameter /
    register_module on an existing name after this call would leave a stale mirror -- call again then.  on=False removes the mirrors."""
    for m in model.modules():
        d = m.__dict__
        for name, child in m._modules.items():
            if on and child is not None:
                d[name] = child
            else:
                d.pop(name, None)
        for name, p in m._parameters.items():
            if on and p is not None:
                d[name] = p
            else:
                d.pop(name, None)
    return model


def bn_tensors(bn):
    """(weight, bias, running_mean, running_var, eps, momentum) of a BatchNorm module, read from the module's own dictionaries:
    `bn.weight` goes through nn.Module.__getattr__ (a Python function that searches three dictionaries), `bn._parameters["weight"]` does not."""
    p, b = bn._parameters, bn._buffers
    return p["weight"], p["bias"], b["running_mean"], b["running_var"], float(bn.eps), float(bn.momentum)


def conv_params(conv):
    """(weight, bias) of a convolution / linear container, without nn.Module.__getattr__ (see bn_tensors)."""
    p = conv._parameters
    return p["weight"], p.get("bias")


_BN_FOLD_CACHE = {}


def bn_eval_scale_shift(bn):
    """Folded eval-mode BatchNorm: y = x*scale + shift.  Cached per module while its parameters / running statistics are
    unchanged (the frozen teacher: computed once instead of 4 small launches per layer per step)."""
    ver = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version, _WEIGHTS_EPOCH[0] if bn.weight.requires_grad else _FROZEN_EPOCH[0],
           bn.weight.data_ptr(), bn.running_var.data_ptr())
    hit = _BN_FOLD_CACHE.get(id(bn))
    if hit is not None and hit[0] == ver and hit[3]() is bn:
        return hit[1], hit[2]
    with torch.no_grad():
        rstd = torch.rsqrt(bn.running_var + bn.eps)
        scale = (bn.weight * rstd).contiguous()
        shift = (bn.bias - bn.running_mean * scale).contiguous()
    _BN_FOLD_CACHE[id(bn)] = (ver, scale, shift, weakref.ref(bn))
    return scale, shift


class _BNEvalActFn(torch.autograd.Function):
    """Eval-mode BN (+residual)(+act) that still backpropagates (student in eval(), rarely needed)."""

    @staticmethod
    def forward(ctx, x, scale, shift, residual, act):
        y = K.affine_act(x, scale, shift, residual, act)
        ctx.act = act
        ctx.has_res = residual is not None
        ctx.save_for_backward(y, scale)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, scale = ctx.saved_tensors
        if ctx.act == 2:
            raise RuntimeError("backward through eval-mode BN+GELU is not implemented")
        g = gy * (y > 0) if ctx.act == 1 else gy
        return g * scale, None, None, (g if ctx.has_res else None), None


def bn_act_eval(x, bn, residual=None, act=1):
    scale, shift = bn_eval_scale_shift(bn)
    if torch.is_grad_enabled() and (x.requires_grad or (residual is not None and residual.requires_grad)):
        return _BNEvalActFn.apply(x, scale, shift, residual, act)
    return K.affine_act(x, scale, shift, residual, act)


class _RowsToDenseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, coords, batch, H, W):
        ctx.save_for_backward(coords)
        ctx.geom = (batch, H, W)
        return K.rows_to_dense(feats, coords, batch, H, W)

    @staticmethod
    def backward(ctx, g):
        (coords,) = ctx.saved_tensors
        b, H, W = ctx.geom
        return K.dense_to_rows(g.contiguous(), coords, b, H, W), None, None, None, None


def rows_to_dense(feats, coords, batch, H, W):
    return _RowsToDenseFn.apply(feats, coords, batch, H, W)


# ------------------------------------------------------------------------------------------ dense map helpers
# A map produced by rows_to_nchw remembers the rows tensor it is a view of, and nchw_to_rows hands that tensor back: consecutive layers
# then pass the SAME (rows) tensor along -- no permute / reshape in the forward, and above all no View / Permute backward nodes
# between two layers' backward functions (module boundaries keep their logical NCHW tensors; the map and its rows share memory and
# version counter, so an in-place change of one is seen by the other).  RD_ROWS_SHORTCUT=0 disables.
ROWS_SHORTCUT = [os.environ.get("RD_ROWS_SHORTCUT", "1") != "0"]


def nchw_to_rows(x):
    """(B,C,H,W) tensor -> (rows (B*H*W, C) view/copy, B, H, W).  Channels-last memory is a free view."""
    B, C, H, W = x.shape
    if ROWS_SHORTCUT[0] and not CAPTURING[0]:          # (a capture works on static tensors that outlive the step)
        r = getattr(x, "_rd_rows", None)
        if r is not None and r.shape[0] == B * H * W and r.shape[1] == C:
            return r, B, H, W
    xr = x.permute(0, 2, 3, 1)
    if not xr.is_contiguous():
        xr = xr.contiguous()
    return xr.reshape(B * H * W, C), B, H, W


def rows_to_nchw(rows, B, H, W):
    """(B*H*W, C) rows -> logical (B,C,H,W) tensor in channels-last memory (no copy)."""
    out = rows.view(B, H, W, rows.shape[1]).permute(0, 3, 1, 2)
    out._rd_rows = rows
    return out


class _Cat2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a_rows, b_rows):
        ctx.c = (a_rows.shape[1], b_rows.shape[1])
        return K.cat2_rows(a_rows, b_rows)

    @staticmethod
    def backward(ctx, g):
        ga, gb = K.split2_rows(g.contiguous(), *ctx.c)
        return ga, gb


def cat_channels(a, b):
    """torch.cat((a, b), dim=1) for two (B, C, H, W) maps in channels-last memory (base_bev_backbone.py:296,
    radar_distill_final.py:121-124): one streaming launch forward, one backward that returns CONTIGUOUS gradients."""
    if not (a.is_cuda and a.dtype == torch.float32 and a.shape[1] % 4 == 0 and b.shape[1] % 4 == 0) or os.environ.get("RD_CAT", "1") == "0":
        return torch.cat((a, b), dim=1)
    ar, B, H, W = nchw_to_rows(a)
    br, _, _, _ = nchw_to_rows(b)
    out = _Cat2Fn.apply(ar, br) if torch.is_grad_enabled() and (a.requires_grad or b.requires_grad) else K.cat2_rows(ar, br)
    return rows_to_nchw(out, B, H, W)


class _DWConvFn(torch.autograd.Function):
    """Depthwise KxK conv (padding K//2) on channels-last rows; weight is the nn.Conv2d parameter [C, 1, K, K]."""

    @staticmethod
    def forward(ctx, x_rows, weight, bias, B, H, W):
        C, K = weight.shape[0], weight.shape[-1]
        note_param_use(weight, bias)
        w_tc = weight.detach().reshape(C, K * K).t().contiguous()
        out = K_.dwconv_fwd(x_rows, w_tc, bias.detach() if bias is not None else None, B, H, W, K)
        ctx.geom = (B, H, W, K)
        ctx.has_bias = bias is not None
        ctx.weight_ref, ctx.bias_ref = weight, bias
        ctx.save_for_backward(x_rows, w_tc)
        return out

    @staticmethod
    def backward(ctx, go):
        x_rows, w_tc = ctx.saved_tensors
        B, H, W, K = ctx.geom
        go = go.contiguous()
        gx = K_.dwconv_fwd(go, w_tc, None, B, H, W, K, flip=True) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            C = x_rows.shape[1]
            # (.t().reshape() alone is a strided VIEW: AccumulateGrad would clone it on the main stream while the side stream still
            # writes it -- found as a wrong dwconv.weight gradient once the host got fast enough to run ahead)
            gw = param_grad_stream(lambda: K_.dwconv_wgrad(x_rows, go, B, H, W, K).t().contiguous().view(C, 1, K, K), x_rows, go, param=ctx.weight_ref)
        gb = param_grad_stream(lambda: K_.colsum(go), go, param=ctx.bias_ref) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb, None, None, None


K_ = K


def dwconv(x_rows, conv, B, H, W):
    return _DWConvFn.apply(x_rows, conv.weight, conv.bias, B, H, W)


class _NConvFn(torch.autograd.Function):
    """Final 3x3 convolutions (64 -> 1..4) of all CenterHead branches at once (nconv.hip)."""

    @staticmethod
    def forward(ctx, y, weight, bias, B, H, W, tab):
        note_param_use(weight, bias)
        out = K.nconv_fwd(y, weight.detach().contiguous(), bias.detach().contiguous() if bias is not None else None, B, H, W, tab)
        ctx.geom, ctx.tab, ctx.has_bias = (B, H, W), tab, bias is not None
        ctx.bias_ref = bias
        ctx.save_for_backward(y, weight)
        return out

    @staticmethod
    def backward(ctx, go):
        y, weight = ctx.saved_tensors
        B, H, W = ctx.geom
        go = go.contiguous()
        gy = K.nconv_dgrad(go, weight.detach().contiguous(), B, H, W, ctx.tab, y.shape[1]) if ctx.needs_input_grad[0] else None
        gw = param_grad_stream(lambda: K.nconv_wgrad(y, go, B, H, W, ctx.tab), y, go, param=weight) if ctx.needs_input_grad[1] else None
        gb = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = param_grad_stream(lambda: K.colsum(go) if go.shape[1] % 4 == 0 else go.sum(0), go, param=ctx.bias_ref)
        return gy, gw, gb, None, None, None, None


_HEAD_FUSED = os.environ.get("RD_HEAD_FUSED", "1") != "0"          # A/B switch of _BNNConvFn


class _BNNConvFn(torch.autograd.Function):
    """Train-mode BatchNorm + ReLU over the batched branch activations followed by the narrow final convolutions, as ONE node: the
    backward never materialises the gradient of the 42 x 64-channel activation tensor (rd_nconv_dgrad_bn recomputes it inside the
    BatchNorm backward's two passes: 1 GB instead of 2.1 GB of traffic for the 352 MB tensor at B = 8)."""

    @staticmethod
    def forward(ctx, raw, gamma, beta, running_mean, running_var, eps, momentum, stats, weight, bias, B, H, W, tab):
        if stats is None:
            stats = K.bn_stats(raw)
        note_param_use(weight, bias)
        y, side = K.bn_train_fwd(raw, stats, gamma, beta, eps, momentum, running_mean, running_var, None, 1)
        out = K.nconv_fwd(y, weight.detach().contiguous(), bias.detach().contiguous() if bias is not None else None, B, H, W, tab)
        ctx.geom, ctx.tab, ctx.bias_ref = (B, H, W), tab, bias
        ctx.save_for_backward(raw, y, gamma, side, weight)
        return out

    @staticmethod
    def backward(ctx, go):
        raw, y, gamma, side, weight = ctx.saved_tensors
        B, H, W = ctx.geom
        go = go.contiguous()
        n = ctx.needs_input_grad
        graw, gg, gb_bn = K.nconv_dgrad_bn(go, weight.detach().contiguous(), raw, gamma, side, B, H, W, ctx.tab)
        gw = param_grad_stream(lambda: K.nconv_wgrad(y, go, B, H, W, ctx.tab), y, go, param=weight) if n[8] else None
        gb = None
        if ctx.bias_ref is not None and n[9]:
            gb = param_grad_stream(lambda: K.colsum(go) if go.shape[1] % 4 == 0 else go.sum(0), go, param=ctx.bias_ref)
        return graw, gg, gb_bn, None, None, None, None, None, gw, gb, None, None, None, None


def bn_relu_nconv_train(raw, gamma, beta, running_mean, running_var, eps, momentum, stats, modules, weight, bias, B, H, W, tab):
    """Fused form of bn_act_train_tensors(act = 1) + nconv for the batched CenterHead branches; falls back to the two nodes when the
    BatchNorm is synchronised over a process group, in deterministic mode (the fused backward uses atomics) or under graph capture."""
    group = sync_group(modules[0] if modules else None)
    if group is not None or K.get_deterministic() or CAPTURING[0] or not raw.is_cuda or raw.shape[1] != tab.nb * 64 or not _HEAD_FUSED:
        y = bn_act_train_tensors(raw, gamma, beta, running_mean, running_var, eps, momentum, act=1, stats=stats, modules=modules)
        return nconv(y, weight, bias, B, H, W, tab)
    if raw.shape[0] <= 1:
        raise ValueError("Expected more than 1 value per channel when training")
    _BN_TOUCHED.extend(modules)
    return _BNNConvFn.apply(raw, gamma, beta, running_mean, running_var, eps, momentum, stats, weight, bias, B, H, W, tab)


def nconv(y, weight, bias, B, H, W, tab):
    if torch.is_grad_enabled() and (y.requires_grad or weight.requires_grad):
        return _NConvFn.apply(y, weight, bias, B, H, W, tab)
    return K.nconv_fwd(y, weight.detach().contiguous(), bias.detach().contiguous() if bias is not None else None, B, H, W, tab)


class _GeluGRNFn(torch.autograd.Function):
    """gelu followed by Global Response Normalisation over each sample's rows (convnext.hip)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, B):
        g1, b1 = gamma.detach().reshape(-1).contiguous(), beta.detach().reshape(-1).contiguous()
        out, a, ssq = K.gelu_grn_fwd(z, B, g1, b1)
        ctx.B = B
        ctx.shapes = (gamma.shape, beta.shape)
        ctx.save_for_backward(z, a, ssq, g1)
        return out

    @staticmethod
    def backward(ctx, go):
        z, a, ssq, g1 = ctx.saved_tensors
        gz, gg, gb = K.gelu_grn_bwd(go.contiguous(), a, z, ssq, ctx.B, g1)
        return gz, gg.reshape(ctx.shapes[0]), gb.reshape(ctx.shapes[1]), None


def gelu_grn(z_rows, grn, B):
    """act = GELU then grn (Basicblock_convn.GRN parameter container) on rows (B*hw, C)."""
    return _GeluGRNFn.apply(z_rows, grn.gamma, grn.beta, B)


class _LayerNormFn(torch.autograd.Function):
    """LayerNorm over the channels of channels-last rows (layernorm.hip)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        y, stat = K.layernorm_fwd(x, g, b, eps)
        ctx.save_for_backward(x, g, stat)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, g, stat = ctx.saved_tensors
        gx, gg, gb = K.layernorm_bwd(x, gy.contiguous(), g, stat)
        return gx, gg, gb, None


def layer_norm_rows(x_rows, weight, bias, eps):
    return _LayerNormFn.apply(x_rows, weight, bias, float(eps))


class _CenterLossFn(torch.autograd.Function):
    """All CenterHead loss terms of all task heads (centerloss.hip): returns (total loss (1,), per-head [hm, loc, iou, iou_reg] (nh, 4))."""

    @staticmethod
    def forward(ctx, maps, cfg, heatmaps, inds, masks, target_boxes, gt_box):
        out, scale, ws = K.center_loss_fwd(cfg, maps, heatmaps, inds, masks, target_boxes, gt_box)
        ctx.cfg = cfg
        ctx.save_for_backward(maps, heatmaps, inds, masks, scale, ws)
        nh = cfg.n_heads
        per_head = out[:4 * nh].view(nh, 4)
        ctx.mark_non_differentiable(per_head)
        return out[4 * nh:], per_head

    @staticmethod
    def backward(ctx, g_total, _g_per_head):
        maps, heatmaps, inds, masks, scale, ws = ctx.saved_tensors
        g = g_total.reshape(1).float().contiguous()
        return K.center_loss_bwd(ctx.cfg, maps, heatmaps, inds, masks, scale, ws, g), None, None, None, None, None, None


def center_loss(maps, cfg, heatmaps, inds, masks, target_boxes, gt_box):
    return _CenterLossFn.apply(maps, cfg, heatmaps, inds, masks, target_boxes, gt_box)
