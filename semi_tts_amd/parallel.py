"""Utterance-level data parallelism: one process per GPU, torch.distributed over RCCL (backend
"nccl" on ROCm) or gloo on CPU.

Inference (the headline benchmark, gen_specgram): utterances are independent, so a batch is
split into contiguous ranges per rank and there is NO data-path collective ("replicas").
Training (BASELINE config 4): every rank runs the same step on its shard; gradients are summed
with a bucketed all-reduce (flat fp32 buckets, default 32 MiB: on the 8-GPU xGMI full mesh a
125 MB payload is ~4 buckets, large enough to run at link rate and few enough to keep launch
overhead negligible), then every rank applies the same clip + optimizer step
(ref: BaseSolver.backward src/solver.py:138-151 needs the GLOBAL grad norm, which is local
after the all-reduce).
"""
import torch
import torch.distributed as dist


def shard_range(n_items, world_size, rank):
    """contiguous, balanced [lo, hi) of `n_items` utterances for `rank` (first ranks get the remainder)"""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_world():
    """(rank, world_size) of this process; (0, 1) outside torch.distributed"""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def max_over_ranks(seconds, device=None):
    """the slowest rank's wall time (what bench.py reports)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


_GRAD_ALLREDUCE = True
_COUNTS = {'grad_buckets': 0, 'syncbn_fwd': 0, 'syncbn_bwd': 0}


def set_gradient_allreduce(enabled=True):
    """measurement hook (bench.py --workload train): switch the gradient all-reduce off to time the step without it"""
    global _GRAD_ALLREDUCE
    _GRAD_ALLREDUCE = bool(enabled)


def collective_counts(reset=False):
    """collectives issued by the LAST training step, by kind (gradient buckets, SyncBN forward / backward)"""
    out = dict(_COUNTS)
    if reset:
        for k in _COUNTS:
            _COUNTS[k] = 0
    return out


def allreduce_gradients(params, bucket_bytes=32 << 20, average=True):
    """Sum (or average) .grad of `params` across ranks with flat buckets; returns the number of
    collectives issued.  Parameters without a gradient contribute zeros so every rank issues the same
    sequence of collectives."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1 or not _GRAD_ALLREDUCE:
        return 0
    world = dist.get_world_size()
    params = [p for p in params if p.requires_grad]
    n_coll, bucket, size = 0, [], 0

    def flush():
        nonlocal n_coll, bucket, size
        if not bucket:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        all_reduce_sum_(flat)
        if average:
            flat /= world
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
        n_coll += 1
        _COUNTS['grad_buckets'] += 1
        bucket, size = [], 0

    for p in params:
        bucket.append(p)
        size += p.numel() * p.element_size()
        if size >= bucket_bytes:
            flush()
    flush()
    return n_coll


class GradReducer:
    """Gradient all-reduce overlapped with the backward pass, without copies.

    The parameters (in REVERSE registration order, which is roughly the order their gradients become ready: postnet first,
    then the decoder's BPTT, then the encoder) are laid out in persistent flat fp32 buckets and every `p.grad` is a VIEW into
    its bucket, so autograd accumulates straight into the communication buffer.  A post-accumulate hook counts the gradients
    of a bucket; when the last one has arrived the bucket's all-reduce is issued asynchronously (RCCL runs it on its own
    stream while the rest of the backward pass keeps the compute stream busy); `finish()` issues what is left, waits, and
    averages.  ~32 MiB buckets: 4 collectives for the 125 MB model -- few, large messages are what the per-link-bound xGMI
    ring wants (ref: the step this replaces is BaseSolver.backward, src/solver.py:138-151, which needs the GLOBAL norm)."""

    def __init__(self, params, bucket_bytes=32 << 20, average=True):
        self.params = [p for p in list(params)[::-1] if p.requires_grad]
        self.average = average
        self.buckets, self.flats, self.slot = [], [], {}
        cur, size = [], 0
        for p in self.params:
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        for bi, bucket in enumerate(self.buckets):
            flat = torch.zeros(sum(p.numel() for p in bucket), device=bucket[0].device, dtype=torch.float32)
            self.flats.append(flat)
            off = 0
            for p in bucket:
                self.slot[p] = (bi, off)
                off += p.numel()
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self.prepare()

    def _view(self, p):
        bi, off = self.slot[p]
        return self.flats[bi][off:off + p.numel()].view_as(p)

    def prepare(self):
        """before every backward pass (after optimizer.zero_grad): zero the buckets and (re-)attach the gradient views"""
        for flat in self.flats:
            flat.zero_()
        for p in self.params:
            p.grad = self._view(p)
        self.count = [0] * len(self.buckets)
        self.fired = set()
        self.works = [None] * len(self.buckets)
        self.launched = [False] * len(self.buckets)

    def _on_grad(self, p):
        if p not in self.slot or p in self.fired:
            return
        if p.grad is None or p.grad.data_ptr() != self._view(p).data_ptr():      # somebody replaced the view: copy back in
            self._view(p).copy_(p.grad)
            p.grad = self._view(p)
        self.fired.add(p)
        bi = self.slot[p][0]
        self.count[bi] += 1
        if self.count[bi] == len(self.buckets[bi]):
            self._launch(bi)

    def _launch(self, bi):
        self.launched[bi] = True
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1 or not _GRAD_ALLREDUCE:
            return
        flat = self.flats[bi]
        _COUNTS['grad_buckets'] += 1
        if flat.is_cuda and dist.get_backend() == 'gloo':      # functional tests on a box with fewer GPUs than ranks
            all_reduce_sum_(flat)
        else:
            self.works[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def finish(self):
        """after backward: reduce the buckets that never filled up (parameters without a gradient this step contribute zeros
        on every rank alike), wait, average; parameters that received no gradient on this rank get `.grad = None` back, as
        autograd would have left them (the optimiser then skips them like the reference's does)"""
        for bi in range(len(self.buckets)):
            if not self.launched[bi]:
                self._launch(bi)
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        for bi, w in enumerate(self.works):
            if w is not None:
                w.wait()
        if world > 1 and self.average and _GRAD_ALLREDUCE:
            for flat in self.flats:
                flat.mul_(1.0 / world)
        for p in self.params:
            if p not in self.fired:
                p.grad = None
        return sum(1 for l in self.launched if l)

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


_SHARD_ROWS = None


def set_shard_rows(mine=None, total=None):
    """utterances of this rank / of the global batch, when the ranks hold different numbers of them (default: equal shards).
    SyncBN needs the global row count on the host without a device round trip."""
    global _SHARD_ROWS
    _SHARD_ROWS = (int(mine), int(total)) if mine else None


def global_rows(local_rows):
    """rows of the global batch a BatchNorm layer sees, from this rank's rows"""
    if not (dist.is_available() and dist.is_initialized()):
        return local_rows
    if _SHARD_ROWS is None:
        return local_rows * dist.get_world_size()
    mine, total = _SHARD_ROWS
    return local_rows * total // mine


def all_gather_(t):
    """(world, *t.shape) tensor of every rank's `t`; device tensors are staged through the host for the gloo backend"""
    world = dist.get_world_size()
    if t.is_cuda and dist.get_backend() == 'gloo':
        h = t.detach().cpu()
        outs = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(outs, h)
        return torch.stack(outs).to(t.device)
    out = torch.empty((world,) + tuple(t.shape), device=t.device, dtype=t.dtype)
    dist.all_gather_into_tensor(out, t.contiguous())
    return out


def broadcast_parameters(module, src=0):
    """make every replica start from rank `src`'s weights and buffers"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


# --------------------------------------------------------------------------------------------- synchronised BatchNorm
_SYNC_BN = False


def sync_batchnorm(enabled=True):
    """Make training-mode BatchNorm use the statistics of the GLOBAL batch (all ranks), like the reference's
    single-process full batch.  Per layer and step: ONE all-gather of (mean, M2, count) in forward, merged
    exactly on every rank in rank order (Chan et al.), and one all-reduce in backward (the two sums of the dx formula)."""
    global _SYNC_BN
    _SYNC_BN = bool(enabled)


def sync_bn_active():
    return _SYNC_BN and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def all_reduce_sum_(t):
    """in-place SUM all-reduce of a small tensor; device tensors are staged through the host for the gloo backend"""
    if t.is_cuda and dist.get_backend() == 'gloo':
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
