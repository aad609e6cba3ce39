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
    single-process full batch.  Per layer and step: two small all-reduces in forward (count + weighted
    means, then the centred second moments) and one in backward (the two sums of the dx formula)."""
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
