"""One-process-per-GPU data parallelism helpers: process-group setup from the torchrun environment,
sample sharding, DDP wrapping with the frozen teacher excluded, max-over-ranks timing and the logging reduction.

Reference: tools/train.py:73-81,174-176 (init + DDP wrap), pcdet/utils/common_utils.py:169-211 (init_dist_pytorch),
tools/train_utils/train_utils.py:73-75 (three `average_reduce_value` calls = 6 pickled all-gathers per step; here ONE
all-reduce of a 3-float tensor, and only when something is logged).  Backend "nccl" is RCCL on ROCm; "gloo" is used by the
CPU tests.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def rehearsal():
    """RD_DP_REHEARSE=1: run the whole data-parallel path -- process group, parameter broadcast, bucketed flat-buffer all-reduce on the
    communication stream, presence mask -- in a world of ONE rank.  The collectives then are RCCL's single-rank forms, but every
    stream hand-over, async work handle and bucket launch of the N > 1 path executes on the GPU (a one-GPU box cannot host two RCCL
    ranks: "duplicate GPU")."""
    return os.environ.get("RD_DP_REHEARSE", "0") == "1"


def _active():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or rehearsal())


def init_distributed(backend="nccl", device=None):
    world, rank, local_rank = env_world()
    if (world > 1 or rehearsal()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend=backend, **kw)
    return world, rank, local_rank


def shard_seed(rank, index, base=0):
    """Seed of the `index`-th synthetic batch of rank `rank`: ranks never see the same samples (DistributedSampler role)."""
    return base + 1000 * rank + index


def wrap_ddp(model, local_rank=None):
    """DistributedDataParallel over the trainable parameters only (frozen teacher parameters have requires_grad=False and are
    ignored by the reducer, tools/train.py:174-176)."""
    if not _active():
        return model
    ids = [local_rank] if (local_rank is not None and next(model.parameters()).is_cuda) else None
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids)


def broadcast_parameters(model, src=0):
    """Rank `src`'s parameters and buffers to every rank, coalesced (what DistributedDataParallel does once at construction)."""
    if not _active():
        return
    tensors = [p.data for p in model.parameters()] + [b.data for b in model.buffers()]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for group in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in group])
        dist.broadcast(flat, src)
        off = 0
        for t in group:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
    # the copies above go through `.data` and move no version counter: every cache keyed on (param._version, weights epoch) -- kernel
    # layouts, split operands, the concatenated CenterHead leaves, folded BatchNorm -- must see that the weights changed
    from . import autograd as A
    A.bump_weights_epoch(frozen=True)


def data_parallel(model, optimizer, device_index=None, mode=None):
    """Set up sample-sharded data parallelism for (model, optimizer); returns the module to call in the training loop.

    mode "flat" (default, RD_DDP=flat): no wrapper -- parameters are broadcast once, the fused optimizer packs all gradients into one
    buffer, all-reduces it with a single RCCL call and consumes the averaged gradients from it (FusedAdamOneCycle
    .enable_flat_allreduce).  BatchNorm running statistics stay local; rank 0's are what a checkpoint stores, exactly as under
    DDP's broadcast_buffers (rank 0's buffers are never overwritten there either).
    mode "torch" (RD_DDP=torch): torch.nn.parallel.DistributedDataParallel as in tools/train.py:174-176 of the reference."""
    if not _active():
        return model
    mode = mode or os.environ.get("RD_DDP", "flat")
    if mode == "torch" or not hasattr(optimizer, "enable_flat_allreduce"):
        from . import autograd as A
        A.WGRAD_STREAM[0] = False        # DDP's reducer hooks read every gradient the moment autograd produces it, on the main stream
        A.CONCAT_LEAVES[0] = False       # ... and only gradients that arrive through the parameters' own AccumulateGrad nodes
        return wrap_ddp(model, device_index)
    broadcast_parameters(model, 0)
    optimizer.enable_flat_allreduce()
    return model


class GradBuckets:
    """Bucketed, overlapped gradient exchange over ONE flat fp32 buffer (99.6 MB, bucketed in reverse module order and
    overlapped with the student backward).  The buffer is laid out in parameter order; buckets are contiguous parameter ranges of
    ~`bucket_bytes`, built from the LAST parameter backwards -- backward produces the head's gradients first and the VFE's last, so
    bucket 0 (head) is complete first.  `ready(i)` is called once per parameter and backward pass (post-accumulate-grad hook); when
    a bucket's last gradient has arrived, `launch(b)` packs it into its slice and starts its all-reduce (async), while backward
    continues to produce the earlier layers' gradients.  `finish()` launches whatever is still open (parameters that received no
    gradient) and waits for all collectives.  Summation order inside a collective does not depend on the bucketing, so the result
    equals the unbucketed all-reduce bit for bit (tests/test_dist_gloo.py, tests/dist_flat_check.py)."""

    def __init__(self, numels, bucket_bytes=25 << 20):
        n = len(numels)
        self.ranges = []                     # (param_lo, param_hi) in parameter order, listed head-first
        hi, acc = n, 0
        for i in range(n - 1, -1, -1):
            acc += numels[i] * 4
            if acc >= bucket_bytes or i == 0:
                self.ranges.append((i, hi))
                hi, acc = i, 0
        self.bucket_of = [0] * n
        for b, (lo, hi_) in enumerate(self.ranges):
            for i in range(lo, hi_):
                self.bucket_of[i] = b
        self.reset()

    def reset(self):
        self.left = [hi - lo for lo, hi in self.ranges]
        self.launched = [False] * len(self.ranges)
        self.seen = [False] * len(self.bucket_of)
        self.dirty = set()                   # buckets that were launched and then received ANOTHER gradient (see ready)

    def ready(self, i):
        """-> bucket index to launch now, or None.
        A parameter that reports a second time before reset() -- a second backward() before step(), i.e. gradient accumulation, or a
        retry -- changes a gradient its bucket may already have sent: the bucket is marked dirty and open_buckets() hands it out
        again, so that finish packs the ACCUMULATED p.grad and all-reduces the slice once more (the slice then holds the sum over
        ranks of the accumulated gradients: correct, at the price of a second collective for that bucket)."""
        b = self.bucket_of[i]
        if self.seen[i]:
            if self.launched[b]:
                self.dirty.add(b)
            return None
        self.seen[i] = True
        self.left[b] -= 1
        if self.left[b] == 0 and not self.launched[b]:
            self.launched[b] = True
            return b
        return None

    def open_buckets(self):
        """Buckets finish() still has to launch: the ones that never completed (parameters without a gradient) and the dirty ones."""
        out = [b for b, done in enumerate(self.launched) if not done or b in self.dirty]
        for b in out:
            self.launched[b] = True
        self.dirty.clear()
        return out


def max_over_ranks(seconds, device="cpu"):
    if not _active():
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def average_scalars(values, device="cpu"):
    """Mean over ranks of a few python floats with ONE collective (data / forward / batch time of the reference's loop)."""
    if not _active():
        return [float(v) for v in values]
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return (t / dist.get_world_size()).tolist()


def barrier():
    if _active():
        dist.barrier()
