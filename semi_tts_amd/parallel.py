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
    """Gradient all-reduce overlapped with the backward pass; gradients are BORN in their communication buffer.

    The parameters (in REVERSE registration order, which is roughly the order their gradients become ready: postnet first,
    then the decoder's BPTT, then the encoder) are laid out in persistent flat fp32 buckets.  Before a backward pass every
    `p.grad` is None and NOTHING is zeroed.  The kernels that produce parameter gradients ask for the parameter's bucket slot
    (`ops.grad_slot(param)` -> this object's `claim`) and write their result there; autograd's AccumulateGrad then ADOPTS that
    view as `p.grad` (an undefined .grad takes the incoming tensor, no accumulate launch).  Gradients that were produced
    elsewhere (small tensors whose producers do not ask; sums autograd formed itself) are copied into their slots by ONE
    multi-tensor launch per bucket when the bucket goes out.  (Round 4 pointed every p.grad at a zeroed slot instead:
    125 MB of fills and one elementwise add launch per parameter -- 144 -- per step.)

    A post-accumulate hook counts the gradients of a bucket; when the last one has arrived the bucket is READY, and ready
    buckets are issued strictly in index order (bucket i only after buckets 0..i-1, as DDP does): ranks whose autograd graphs
    differ (a data-dependent `ignore_speech_cycle`, per-rank `skip_prob` draws) still issue the same collectives in the same
    order.  The all-reduce is asynchronous (RCCL runs it on its own stream while the rest of the backward pass keeps the
    compute stream busy); `finish()` issues what is left and waits.  ~32 MiB buckets: 4 collectives for the 125 MB model -- few,
    large messages are what the per-link-bound xGMI ring wants (ref: the step this replaces is BaseSolver.backward,
    src/solver.py:138-151, which needs the GLOBAL norm).

    average: `finish()` leaves the mean over the ranks in every `.grad` (one multiply launch per bucket); with
    defer_average=True it leaves the SUM and `grad_scale` = 1 / world for the caller to fold into its clip launch
    (optim.clip_grad_norm_(..., pre_scale=...)): no launch at all.

    Which parameters received a gradient on ANY rank travels with the last bucket (one flag per parameter behind its
    gradients, no extra collective): a parameter keeps its reduced gradient on every rank when at least one rank produced
    one, and gets `.grad = None` on every rank when none did -- the optimiser then skips it everywhere or nowhere, and the
    global clip norm is the same on all ranks."""

    def __init__(self, params, bucket_bytes=32 << 20, average=True, static_graph=False, defer_average=False):
        """static_graph=True (a promise, as DistributedDataParallel's): every backward pass produces gradients for the same parameters
        in the same order on every rank.  The first pass runs as usual (one hook per parameter, registration-order buckets) and
        records the order in which the gradients arrive; the buckets are then REBUILT in that order over the parameters that
        actually receive gradients -- a bucket becomes ready as early as its position in the backward pass allows (the decoder's
        weight gradients go out while the encoder's backward still runs; parameters the step never touches, e.g. the speech
        encoder in the paired TTS step, no longer hold the last bucket back until finish()) -- and only the hook of each bucket's
        LAST parameter stays registered: one Python call per bucket and step instead of one per parameter.  The promise is CHECKED:
        a bucket whose closing gradient arrives while another of its gradients is still missing, a gradient produced after its
        bucket went out, or a gradient for a parameter the recorded pass never touched (a teacher-forcing schedule that starts
        feeding outputs back, a parameter that gets unfrozen) is noticed in the same step -- that step's buckets then go out from
        finish(), complete, and the reducer returns to the per-parameter form for good, with a warning.  Trainers whose graph
        depends on the data on purpose (the speech / text cycles) leave it off."""
        self.all_params = [p for p in list(params)[::-1] if p.requires_grad]
        self.params = list(self.all_params)
        self.dropped = []                              # static graph: parameters the recorded pass never touched
        self.average = average
        self.defer_average = bool(defer_average)
        self.grad_scale = 1.0                          # what the gradients must still be multiplied by (defer_average)
        self.bucket_bytes = int(bucket_bytes)
        self.static_graph = bool(static_graph)
        self._sparse = False                           # static_graph: True once the buckets follow the recorded order
        self._fire_order = []                          # the recorded pass: parameters in the order their gradients arrived
        self._hooks = {}
        self._active = False
        self._ctl = None                               # static graph under torch.distributed: the ranks' agreement about a step (finish / _settle)
        self._ctl_pending = []                         # agreements not yet acted on: (kind, value or ring slot, world, recorded order, steps waited)
        self._ctl_used = False
        self._recorded_order = None
        self.stats = {'born_in_slot': 0, 'gathered': 0, 'zeroed': 0}      # of the last step (tests, bench)
        self._layout()
        self._hooks = {p: p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params}
        from . import ops
        ops.set_grad_sink(self)
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
        pad4 = lambda n: (n + 3) // 4 * 4              # every slot starts on a 16-byte boundary: the producing kernels, the clip
        for bi, bucket in enumerate(self.buckets):     # and Adam launches then take the same (16-byte) code paths as on tensors of their own
            n = sum(pad4(p.numel()) for p in bucket)
            if bi == len(self.buckets) - 1:
                n += len(self.params)                  # the "fired on some rank" flags ride behind the last bucket's gradients
            flat = torch.zeros(n, device=bucket[0].device, dtype=torch.float32)
            self.flats.append(flat)
            off = 0
            for p in bucket:
                self.slot[p] = (bi, off)
                off += pad4(p.numel())
        self.flags = self.flats[-1][self.flats[-1].numel() - len(self.params):] if self.flats else None
        self.ptr = {p: self.flats[bi].data_ptr() + 4 * off for p, (bi, off) in self.slot.items()}      # address of every slot
        self._by_ptr = {p.data_ptr(): p for p in self.params}                                        # parameter storage -> parameter

    def _rebuild_static(self):
        """after the recorded pass (its gradients are reduced in the OLD buckets): new buckets in arrival order over the parameters
        that fired, this step's gradients carried over, one hook per bucket"""
        order = self._recorded_order if self._recorded_order is not None else self._fire_order
        old_grad = {p: p.grad for p in order}
        self.dropped = [p for p in self.params if p not in old_grad]
        for h in self._hooks.values():
            h.remove()
        self.params = list(order)
        self._layout()
        for p in self.params:
            if old_grad[p] is not None:                # (decided at the next step's prepare(): the gradients are used up and gone by then)
                v = self._view(p)
                v.copy_(old_grad[p])
                p.grad = v
        for p in self.dropped:
            p.grad = None                              # never touched by this (static) step: the optimiser skips it on every rank
        closers = [bucket[-1] for bucket in self.buckets]
        self._hooks = {p: p.register_post_accumulate_grad_hook(self._on_grad) for p in closers}
        self._sparse = True

    def _revert_dynamic(self, why):
        """the static promise did not hold: back to one hook per parameter over ALL parameters (this step's gradients stay where
        they are -- views of the old buckets -- until the optimiser has used them)"""
        import warnings
        warnings.warn('GradReducer: static_graph promise broken (%s); back to per-parameter hooks' % why, RuntimeWarning)
        for h in self._hooks.values():
            h.remove()
        self.static_graph, self._sparse = False, False
        self.params, self.dropped = list(self.all_params), []
        self._layout()
        self._hooks = {p: p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params}

    def _view(self, p):
        bi, off = self.slot[p]
        return self.flats[bi][off:off + p.numel()].view_as(p)

    def prepare(self):
        """before every backward pass: no gradient exists (nothing is zeroed, nothing is attached)"""
        self._settle()                                 # (what the ranks agreed on about the step before: same layout change on every rank)
        for p in self.params:
            if p.grad is not None:
                p.grad = None
        for p in self.dropped:
            if p.grad is not None:
                p.grad = None
        self.count = [0] * len(self.buckets)
        self.fired = set()
        self.claimed = set()
        self._fire_order = []
        self._violated = None
        self.works = [None] * len(self.buckets)
        self.ready = [False] * len(self.buckets)
        self.launched = [False] * len(self.buckets)
        self.next = 0                                  # the next bucket to issue (in-order launches)
        self.order = []                                # launch order of this step (tests)
        self.stats = {'born_in_slot': 0, 'gathered': 0, 'zeroed': 0}
        self._active = True

    def claim(self, param):
        """ops.grad_slot: the bucket slot of `param` for the kernel that is about to produce its gradient (first request of a step
        only; a second producer of the same parameter, or one that comes after the bucket went out, gets None and allocates)"""
        if not self._active:
            return None
        p = self._by_ptr.get(param.data_ptr())
        if p is None or p in self.claimed or p.numel() != param.numel():
            return None
        if self.launched[self.slot[p][0]]:
            self._violated = self._violated or 'a gradient was produced after its bucket had gone out'
            return None
        self.claimed.add(p)
        return self._view(p)

    def _on_grad(self, p):
        if self._sparse:                               # static graph: p completes its bucket (recorded in the first pass)
            bi = self.slot[p][0]
            if not self.ready[bi]:
                self.ready[bi] = True
                self._launch_ready()
            return
        if p not in self.slot or p in self.fired:
            return
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
            if not self._launch(self.next, final=everything):
                break
            self.next += 1

    def _gather(self, bi, final):
        """every gradient of bucket bi into its slot: most were born there; the others are copied in by ONE multi-tensor launch and
        re-pointed at their slots; the slot of a parameter without a gradient is zeroed (only when `final`).  False: a gradient is
        still missing and the bucket must wait (static graph: the promise is broken)"""
        stray, missing = [], []
        ptr = self.ptr
        for p in self.buckets[bi]:
            g = p.grad
            if g is None:
                missing.append(p)
            elif g.data_ptr() != ptr[p]:
                stray.append(p)
        if missing and not final:
            return False
        self.stats['born_in_slot'] += len(self.buckets[bi]) - len(stray) - len(missing)
        self.stats['gathered'] += len(stray)
        self.stats['zeroed'] += len(missing)
        if stray:
            views = [self._view(p) for p in stray]
            srcs = [p.grad for p in stray]
            if views[0].is_cuda and all(g.is_contiguous() and g.dtype == torch.float32 and g.numel() == v.numel() for g, v in zip(srcs, views)):
                from . import ops
                ops.mt_copy(views, srcs)
            else:
                for v, g in zip(views, srcs):
                    v.copy_(g)
            for p, v in zip(stray, views):
                p.grad = v
        for p in missing:
            self._view(p).zero_()
        return True

    def _launch(self, bi, final=False):
        from . import ops
        ops.flush_wgrads()          # (weight-gradient products still queued write into these slots: out before the bucket is read)
        if not self._gather(bi, final):
            self._violated = self._violated or 'a bucket closed while one of its gradients was missing'
            self.ready[bi] = False
            return False
        self.launched[bi] = True
        self.order.append(bi)
        if not dist_on() or not _GRAD_ALLREDUCE:
            return True
        flat = self.flats[bi]
        if bi == len(self.buckets) - 1:                # flags of this rank; final here: every earlier bucket is out already
            has = [p.grad is not None for p in self.params]
            if all(has):
                self.flags.fill_(1.0)
            else:
                self.flags.copy_(torch.tensor([1.0 if h else 0.0 for h in has]), non_blocking=True)
        _COUNTS['grad_buckets'] += 1
        if flat.is_cuda and dist.get_backend() == 'gloo':      # functional tests on a box with fewer GPUs than ranks
            all_reduce_sum_(flat)
        else:
            self.works[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
            _HANDOFF['async_buckets'] += 1
        return True

    def finish(self):
        """after backward: issue the buckets that never filled up (parameters without a gradient this step contribute zeros),
        wait, average (or leave the sums and set `grad_scale`, defer_average).  A parameter that received no gradient on ANY rank
        keeps `.grad = None`, as autograd left it (the optimiser skips it like the reference's does); one that fired on some
        other rank gets the reduced gradient here too."""
        self._launch_ready(everything=True)
        self._active = False
        reduced = dist_on() and _GRAD_ALLREDUCE
        world = dist.get_world_size() if reduced else 1
        for w in self.works:
            if w is not None:
                w.wait()
        silent = [p for p in self.params if p.grad is None]
        if silent:
            # only now, and only in the irregular case, the flags are read back (one small device -> host copy)
            anywhere = self.flags.detach().cpu().tolist() if reduced else None
            for p in silent:
                if anywhere is not None and anywhere[self.index[p]] != 0.0:
                    p.grad = self._view(p)
        extra = [p for p in self.dropped if p.grad is not None]
        if extra:        # static graph: a parameter outside the recorded set fired -- the promise is gone
            self._violated = self._violated or 'a parameter outside the recorded set received a gradient'
        # The promise is per rank, the bucket layout is not: every rank must change it in the same step or the next all-reduces differ in
        # size and order (advisor, round 5).  With a static graph every rank adds (violated, h, h^2) -- h a hash of the order its gradients
        # arrived in during the recorded pass -- to a three-double all-reduce behind the buckets; the result comes to the host on its own
        # (pinned copy + event) and is looked at when the SECOND step after it begins (_settle, CTL_DELAY): nobody waits for the GPU in a
        # regular step.
        # A rank that saw a violation itself reads the sum now: only if EVERY rank saw one (a change of the graph that comes with the step
        # count, e.g. a teacher-forcing schedule) are this step's stray gradients reduced on their own -- a collective the other ranks
        # would not join otherwise.
        together = True
        recording = self.static_graph and not self._sparse and bool(self._fire_order)
        if reduced and (self.static_graph or self._sparse):
            h = float(self._order_hash()) if recording else 0.0
            total = self._exchange_control(1.0 if self._violated else 0.0, h, world, wait=bool(self._violated))
            if self._violated:
                together = total is not None and total[0] >= world
        if extra and reduced and together:
            flat = torch.cat([p.grad.reshape(-1) for p in extra])
            all_reduce_sum_(flat)
            off = 0
            for p in extra:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        elif extra and reduced:
            import warnings
            warnings.warn('GradReducer: a gradient outside the recorded set on THIS rank only; it stays unreduced for this step', RuntimeWarning)
        self.grad_scale = 1.0
        if world > 1 and self.average:
            if self.defer_average:
                self.grad_scale = 1.0 / world
            else:
                n_flags = len(self.params)
                for bi, flat in enumerate(self.flats):
                    (flat[:flat.numel() - n_flags] if bi == len(self.flats) - 1 else flat).mul_(1.0 / world)
                for p in extra:
                    p.grad.mul_(1.0 / world)
        issued = sum(1 for l in self.launched if l)
        if self._ctl_used:
            self._ctl_used = False
            self._ctl_pending[-1][3] = list(self._fire_order) if recording else None
        else:                                          # no exchange (one process, or collectives off): decide here and now
            self._recorded_order = list(self._fire_order) if recording else None
            self._decide(bool(self._violated), True, self._violated)
        return issued

    def _order_hash(self):
        """the recorded arrival order as a number < 2^20 (its square is exact in a double)"""
        h = 0
        for p in self._fire_order:
            h = (h * 1000003 + self.index[p] + 1) % 1048573
        return h

    def _exchange_control(self, violated, h, world, wait):
        """all-reduce (violated, h, h^2) over the ranks; the sum goes to a pinned host buffer behind an event (read by _settle, or here
        when `wait`).  Returns the summed triple when it was read."""
        dev = self.flats[0].device
        RING = 4
        if self._ctl is None or self._ctl.device != dev:
            self._ctl = torch.zeros(3, device=dev, dtype=torch.float64)
            self._ctl_host = torch.zeros(RING, 3, dtype=torch.float64).pin_memory() if dev.type == 'cuda' else None
            self._ctl_events = [torch.cuda.Event() for _ in range(RING)] if dev.type == 'cuda' else None
            self._ctl_slot = 0
        ctl = self._ctl
        ctl.zero_()
        if violated:
            ctl[0] = violated
        if h:
            ctl[1], ctl[2] = h, h * h
        all_reduce_sum_(ctl)
        self.stats['control'] = self.stats.get('control', 0) + 1
        self._ctl_used = True
        if self._ctl_host is None:                     # CPU tensors (gloo tests): the collective was synchronous
            val = ctl.tolist()
            self._ctl_pending.append(['value', val, world, None, 0])
            return val
        slot = self._ctl_slot
        self._ctl_slot = (slot + 1) % RING
        self._ctl_host[slot].copy_(ctl, non_blocking=True)
        self._ctl_events[slot].record()
        self._ctl_pending.append(['event', slot, world, None, 0])
        if wait:
            self._ctl_events[slot].synchronize()
            return self._ctl_host[slot].tolist()
        return None

    CTL_DELAY = 2        # device tensors: an agreement is acted on when the SECOND step after it begins -- its copy has long landed by then,
                         # so the host, which runs about a step ahead of the GPU, never waits for it (acting one step later cost 0.5 ms per step)

    def _settle(self):
        """at the beginning of a step: act on what the ranks exchanged at the end of an earlier step -- every rank reads the same sums and
        changes (or keeps) its bucket layout in the same step"""
        keep = []
        for ent in self._ctl_pending:
            kind, val, world, order, age = ent
            ent[4] = age = age + 1
            if kind == 'event' and age < self.CTL_DELAY:
                keep.append(ent)
                continue
            if kind == 'event':
                self._ctl_events[val].synchronize()
                val = self._ctl_host[val].tolist()
            viol, h, h2 = val
            same_order = world * h2 == h * h           # (sum h)^2 == world * sum h^2  <=>  every rank recorded the same order; exact in doubles
            why = 'reported by %d of %d ranks' % (int(viol), world) if viol > 0 else 'the ranks recorded different gradient orders'
            self._recorded_order = order
            self._decide(viol > 0, same_order, why)
        self._ctl_pending = keep

    def _decide(self, violated, same_order, why):
        if (violated or not same_order) and (self._sparse or self.static_graph):
            self._revert_dynamic(why)
        elif self.static_graph and not self._sparse and self._recorded_order and all(p in self.slot for p in self._recorded_order):
            self._rebuild_static()
        self._recorded_order = None

    def close(self):
        for h in self._hooks.values():
            h.remove()
        self._hooks = {}
        self._active = False
        from . import ops
        ops.set_grad_sink(None, only_if=self)


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
