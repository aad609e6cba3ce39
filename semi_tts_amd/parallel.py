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


_FORCE = False


def force_collectives(enabled=True):
    """Issue every collective even in a world of ONE rank (bench.py --dist, tests/test_gpu_rccl.py): a single-GPU box can then
    execute the RCCL branch of the reducer, of SyncBN and of the timing helpers end to end -- the sums over one rank are the
    identity, so the step must reproduce the non-distributed one."""
    global _FORCE
    _FORCE = bool(enabled)


def dist_on():
    """torch.distributed is initialised and there is something to reduce over (or collectives are forced)"""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


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
    if not dist_on():
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


_GRAD_ALLREDUCE = True
_COUNTS = {'grad_buckets': 0, 'syncbn_fwd': 0, 'syncbn_bwd': 0}
_HANDOFF = {'async_buckets': 0}      # buckets issued with async_op=True (the RCCL branch), since the last collective_counts(reset=True)


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
        _HANDOFF['async_buckets'] = 0
    return out


def async_bucket_count():
    """gradient buckets of the last step that went out as asynchronous collectives on the backend's own stream (RCCL)"""
    return _HANDOFF['async_buckets']


def allreduce_gradients(params, bucket_bytes=32 << 20, average=True):
    """Sum (or average) .grad of `params` across ranks with flat buckets; returns the number of
    collectives issued.  Parameters without a gradient contribute zeros so every rank issues the same
    sequence of collectives."""
    if not dist_on() or not _GRAD_ALLREDUCE:
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
    of a bucket; when the last one has arrived the bucket is READY, and ready buckets are issued strictly in index order
    (bucket i only after buckets 0..i-1, as DDP does): ranks whose autograd graphs differ (a data-dependent
    `ignore_speech_cycle`, per-rank `skip_prob` draws) still issue the same collectives in the same order.  The all-reduce is
    asynchronous (RCCL runs it on its own stream while the rest of the backward pass keeps the compute stream busy);
    `finish()` issues what is left, waits, and averages.  ~32 MiB buckets: 4 collectives for the 125 MB model -- few, large
    messages are what the per-link-bound xGMI ring wants (ref: the step this replaces is BaseSolver.backward,
    src/solver.py:138-151, which needs the GLOBAL norm).

    Which parameters received a gradient on ANY rank travels with the last bucket (one flag per parameter behind its
    gradients, no extra collective): a parameter keeps its averaged gradient on every rank when at least one rank produced
    one, and gets `.grad = None` on every rank when none did -- the optimiser then skips it everywhere or nowhere, and the
    global clip norm is the same on all ranks."""

    def __init__(self, params, bucket_bytes=32 << 20, average=True, static_graph=False):
        """static_graph=True (a promise, as DistributedDataParallel's): every backward pass produces gradients for the same parameters
        in the same order on every rank.  The first pass runs as usual (one hook per parameter, registration-order buckets) and
        records the order in which the gradients arrive; the buckets are then REBUILT in that order over the parameters that
        actually receive gradients -- a bucket becomes ready as early as its position in the backward pass allows (the decoder's
        weight gradients go out while the encoder's backward still runs; parameters the step never touches, e.g. the speech
        encoder in the paired TTS step, no longer hold the last bucket back until finish()) -- and only the hook of each bucket's
        LAST parameter stays registered: one Python call per bucket and step instead of one per parameter (144 calls, ~0.4 ms of
        host time per step for the paired TTS model).  Trainers whose graph depends on the data (the speech / text cycles:
        `ignore_speech_cycle`, per-rank `skip_prob` draws) must leave it off."""
        self.params = [p for p in list(params)[::-1] if p.requires_grad]
        self.average = average
        self.bucket_bytes = int(bucket_bytes)
        self.static_graph = bool(static_graph)
        self._sparse = False                           # static_graph: True once the buckets follow the recorded order
        self._fire_order = []                          # the recorded pass: parameters in the order their gradients arrived
        self._hooks = {}
        self._layout()
        self._hooks = {p: p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params}
        self.prepare()

    def _layout(self):
        """flat buckets over self.params (in that order) + the gradient views' slots"""
        self.buckets, self.flats, self.slot = [], [], {}
        cur, size = [], 0
        for p in self.params:
            cur.append(p)
            size += p.numel() * 4
            if size >= self.bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self.index = {p: i for i, p in enumerate(self.params)}
        for bi, bucket in enumerate(self.buckets):
            n = sum(p.numel() for p in bucket)
            if bi == len(self.buckets) - 1:
                n += len(self.params)                  # the "fired on some rank" flags ride behind the last bucket's gradients
            flat = torch.zeros(n, device=bucket[0].device, dtype=torch.float32)
            self.flats.append(flat)
            off = 0
            for p in bucket:
                self.slot[p] = (bi, off)
                off += p.numel()
        self.flags = self.flats[-1][self.flats[-1].numel() - len(self.params):] if self.flats else None

    def _rebuild_static(self):
        """after the recorded pass (its gradients are reduced and averaged in the OLD buckets): new buckets in arrival order over
        the parameters that fired, this step's gradients carried over, one hook per bucket"""
        old_view = {p: self._view(p) for p in self._fire_order}
        dropped = [p for p in self.params if p not in old_view]
        for h in self._hooks.values():
            h.remove()
        self.params = list(self._fire_order)
        self._layout()
        for p in self.params:
            v = self._view(p)
            v.copy_(old_view[p])
            p.grad = v
        for p in dropped:
            p.grad = None                              # never touched by this (static) step: the optimiser skips it on every rank
        closers = [bucket[-1] for bucket in self.buckets]
        self._hooks = {p: p.register_post_accumulate_grad_hook(self._on_grad) for p in closers}
        self._sparse = True

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
        self._fire_order = []
        self.works = [None] * len(self.buckets)
        self.ready = [False] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        self.next = 0                                  # the next bucket to issue (in-order launches)
        self.order = []                                # launch order of this step (tests)

    def _on_grad(self, p):
        if self._sparse:                               # static graph: p completes its bucket (recorded in the first pass)
            bi = self.slot[p][0]
            if not self.ready[bi]:
                self.ready[bi] = True
                self._launch_ready()
            return
        if p not in self.slot or p in self.fired:
            return
        if p.grad is None or p.grad.data_ptr() != self._view(p).data_ptr():      # somebody replaced the view: copy back in
            self._view(p).copy_(p.grad)
            p.grad = self._view(p)
        self.fired.add(p)
        if self.static_graph:
            self._fire_order.append(p)
        bi = self.slot[p][0]
        self.count[bi] += 1
        if self.count[bi] == len(self.buckets[bi]):
            self.ready[bi] = True
            self._launch_ready()

    def _launch_ready(self, everything=False):
        while self.next < len(self.buckets) and (everything or self.ready[self.next]):
            self._launch(self.next)
            self.next += 1

    def _launch(self, bi):
        self.launched[bi] = True
        self.order.append(bi)
        if not dist_on() or not _GRAD_ALLREDUCE:
            return
        flat = self.flats[bi]
        if bi == len(self.buckets) - 1 and not self._sparse:    # flags of this rank; final here: every earlier bucket is out already
            if len(self.fired) == len(self.params):
                self.flags.fill_(1.0)
            else:
                self.flags.copy_(torch.tensor([1.0 if p in self.fired else 0.0 for p in self.params]), non_blocking=True)
        _COUNTS['grad_buckets'] += 1
        if bi == len(self.buckets) - 1 and self._sparse:
            self.flags.fill_(1.0)                      # (static graph: every parameter fires every step)
        if flat.is_cuda and dist.get_backend() == 'gloo':      # functional tests on a box with fewer GPUs than ranks
            all_reduce_sum_(flat)
        else:
            self.works[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
            _HANDOFF['async_buckets'] += 1

    def finish(self):
        """after backward: issue the buckets that never filled up (parameters without a gradient this step contribute zeros),
        wait, average.  A parameter that received no gradient on ANY rank gets `.grad = None` back, as autograd would have
        left it (the optimiser skips it like the reference's does); one that fired on some other rank keeps the averaged
        gradient here too."""
        self._launch_ready(everything=True)
        reduced = dist_on() and _GRAD_ALLREDUCE
        world = dist.get_world_size() if reduced else 1
        for w in self.works:
            if w is not None:
                w.wait()
        if not self._sparse and len(self.fired) < len(self.params):
            # only now, and only in the irregular case, the flags are read back (one small device -> host copy)
            anywhere = self.flags.detach().cpu().tolist() if reduced else None
            for i, p in enumerate(self.params):
                if p not in self.fired and (anywhere is None or anywhere[i] == 0.0):
                    p.grad = None
        if world > 1 and self.average:
            n_flags = len(self.params)
            for bi, flat in enumerate(self.flats):
                (flat[:flat.numel() - n_flags] if bi == len(self.flats) - 1 else flat).mul_(1.0 / world)
        issued = sum(1 for l in self.launched if l)
        if self.static_graph and not self._sparse and self._fire_order:
            self._rebuild_static()
        return issued

    def close(self):
        for h in self._hooks.values():
            h.remove()
        self._hooks = {}


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
    if not dist_on():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)
        # the write went through .data, behind autograd's back: bump the version counter, which the cached weight layouts
        # (ops._ParamLayouts, the packed VQ table of L2Embedding) are keyed on
        torch.autograd.graph.increment_version(t)


# --------------------------------------------------------------------------------------------- synchronised BatchNorm
_SYNC_BN = False


def sync_batchnorm(enabled=True):
    """Make training-mode BatchNorm use the statistics of the GLOBAL batch (all ranks), like the reference's
    single-process full batch.  Per layer and step: ONE all-gather of (mean, M2, count) in forward, merged
    exactly on every rank in rank order (Chan et al.), and one all-reduce in backward (the two sums of the dx formula)."""
    global _SYNC_BN
    _SYNC_BN = bool(enabled)


def sync_bn_active():
    return _SYNC_BN and dist_on()


def all_reduce_sum_(t):
    """in-place SUM all-reduce of a small tensor; device tensors are staged through the host for the gloo backend"""
    if t.is_cuda and dist.get_backend() == 'gloo':
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
