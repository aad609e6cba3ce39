"""Data-parallel training on the HIP path: two processes (one GPU shared, gloo for the tiny collectives so that the test
runs on a 1-GPU box) each take half of a batch; with synchronised BatchNorm statistics and the bucketed gradient
all-reduce the averaged gradients must equal the single-process full-batch gradients.  Needs a real MI355X."""
import os

import pytest
import torch

from helpers import maxdiff

pytestmark = pytest.mark.gpu


def _build(dev):
    from conftest import load_golden
    from helpers import tiny_tacotron
    W, A, meta = load_golden('tts_tiny_train_tf')
    m = tiny_tacotron(meta, W, dev).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.decoder.prenet_dropout = 0.0
    m.decoder.prenet.apply_dropout = 0.0
    return m, A, meta


def _step(m, A, rows, dev, weight=1.0):
    from semi_tts_amd import autograd as AG
    txt, spk, teacher = (A[k][rows].to(dev) for k in ('txt_embed', 'spkr_embed', 'teacher'))
    lin_t = torch.rand(4, teacher.shape[1], 20, generator=torch.Generator().manual_seed(3))[rows].to(dev)
    mel, lin, _, _ = m(txt, None, teacher, spk, tf_rate=1.0)
    loss = (AG.freq_loss(mel, teacher, 22050, 8) + AG.freq_loss(lin, lin_t, 22050, 8)) * weight
    loss.backward()
    return float(loss.detach())


def _worker(rank, world, port, out_path, cut=2):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from semi_tts_amd import parallel
    dev = torch.device('cuda:0')
    m, A, _ = _build(dev)
    parallel.sync_batchnorm(True)
    parallel.broadcast_parameters(m)
    # rank 0 takes utterances [0, cut), rank 1 [cut, 4); each local mean loss is weighted by its share of the global batch
    # (n_r * world / N) so that the AVERAGE over ranks is the global mean loss also for unequal shards
    rows = slice(0, cut) if rank == 0 else slice(cut, 4)
    n_r = cut if rank == 0 else 4 - cut
    red = parallel.GradReducer(m.parameters(), bucket_bytes=64 << 10)               # small buckets: several collectives,
    loss = _step(m, A, rows, dev, weight=n_r * world / 4.0)                         # issued from autograd hooks during backward
    n_coll = red.finish()
    assert n_coll > 1
    assert parallel.collective_counts()['syncbn_fwd'] == 6 and parallel.collective_counts()['syncbn_bwd'] == 6   # one each per BN layer; the 8 layers of the conv bank share theirs
    if rank == 0:
        torch.save({'loss': loss, 'grads': {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None},
                    'stats': {k: v.cpu() for k, v in m.state_dict().items() if 'running_' in k}}, out_path)
    dist.destroy_process_group()


@pytest.mark.parametrize('cut', [2, 3])        # equal shards (2 + 2 utterances) and unequal ones (3 + 1)
def test_two_rank_sync_bn_training_equals_single_process(tmp_path, cut):
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    dev = torch.device('cuda:0')
    m, A, _ = _build(dev)
    _step(m, A, slice(0, 4), dev)
    ref = {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None}
    ref_stats = {k: v.cpu() for k, v in m.state_dict().items() if 'running_' in k}
    out = str(tmp_path / 'dp.pt')
    ctx = mp.start_processes(_worker, args=(2, 29517 + os.getpid() % 500 + 7 * cut, out, cut), nprocs=2, join=False, start_method='spawn')
    import time
    deadline = time.time() + 90                 # never hang the suite: a stuck rank is killed and the test fails
    while not ctx.join(timeout=5):
        if time.time() > deadline:
            for p in ctx.processes:
                p.kill()
            pytest.fail('data-parallel workers did not finish within 90 s')
    got = torch.load(out)
    worst = 0.0
    for k, g in ref.items():
        scale = float(g.abs().max())
        if scale < 1e-6:           # analytically zero (a bias in front of a BatchNorm): fp32 round-off on both sides
            assert maxdiff(got['grads'][k], g) < 1e-6, k
            continue
        e = maxdiff(got['grads'][k], g) / scale
        worst = max(worst, e)
        assert e < 5e-5, (k, e)          # same arithmetic split differently over two ranks (fp32 summation order)
    for k, v in ref_stats.items():
        assert maxdiff(got['stats'][k], v) < 1e-5, k
    print('worst relative gradient difference DP vs single process: %.2e' % worst)


def _batch12(A):
    """12 utterances of the tiny model's shapes (the golden holds 4): seeded, the same in every process"""
    g = torch.Generator().manual_seed(12)
    txt = torch.randn(12, *A['txt_embed'].shape[1:], generator=g) * 0.5
    spk = torch.randn(12, *A['spkr_embed'].shape[1:], generator=g) * 0.5
    teacher = torch.rand(12, *A['teacher'].shape[1:], generator=g)
    lin_t = torch.rand(12, A['teacher'].shape[1], 20, generator=g)
    return txt, spk, teacher, lin_t


def _step12(m, batch, rows, dev, weight=1.0):
    from semi_tts_amd import autograd as AG
    txt, spk, teacher, lin_t = (t[rows].to(dev) for t in batch)
    mel, lin, _, _ = m(txt, None, teacher, spk, tf_rate=1.0)
    loss = (AG.freq_loss(mel, teacher, 22050, 8) + AG.freq_loss(lin, lin_t, 22050, 8)) * weight
    loss.backward()
    return float(loss.detach())


_SIZES8 = [3, 2, 2, 1, 1, 1, 1, 1]


def _worker8(rank, world, port, out_path):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from semi_tts_amd import parallel
    dev = torch.device('cuda:0')
    m, A, _ = _build(dev)
    batch = _batch12(A)
    parallel.sync_batchnorm(True)
    parallel.broadcast_parameters(m)
    lo = sum(_SIZES8[:rank])
    n_r = _SIZES8[rank]
    red = parallel.GradReducer(m.parameters(), bucket_bytes=64 << 10, average=True, defer_average=True)
    loss = _step12(m, batch, slice(lo, lo + n_r), dev, weight=n_r * world / 12.0)
    n_coll = red.finish()
    assert n_coll > 1 and red.grad_scale == 1.0 / world
    c = parallel.collective_counts()
    assert c['syncbn_fwd'] == 6 and c['syncbn_bwd'] == 6
    if rank == 0:
        torch.save({'grads': {k: (p.grad * red.grad_scale).cpu() for k, p in m.named_parameters() if p.grad is not None},
                    'stats': {k: v.cpu() for k, v in m.state_dict().items() if 'running_' in k}}, out_path)
    dist.destroy_process_group()


def test_eight_rank_sync_bn_training_with_unequal_shards_equals_single_process(tmp_path):
    """Eight ranks before an 8-GPU box sees them (here: eight processes sharing the test GPU, gloo for the collectives): shards of 3 / 2 / 2 /
    1 / 1 / 1 / 1 / 1 utterances of a batch of 12, SyncBN statistics merged over the ranks' true row counts, gradients born in their buckets,
    the sums all-reduced and 1 / world applied afterwards: averaged gradients and running statistics = the single-process full batch."""
    import time
    import torch.multiprocessing as mp
    dev = torch.device('cuda:0')
    m, A, _ = _build(dev)
    _step12(m, _batch12(A), slice(0, 12), dev)
    ref = {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None}
    ref_stats = {k: v.cpu() for k, v in m.state_dict().items() if 'running_' in k}
    out = str(tmp_path / 'dp8.pt')
    import socket
    from torch.multiprocessing.spawn import ProcessExitedException
    for attempt in range(3):
        with socket.socket() as sk:                 # a port nobody holds (fixed offsets collided once with a lingering listener)
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        ctx = mp.start_processes(_worker8, args=(8, port, out), nprocs=8, join=False, start_method='spawn')
        deadline = time.time() + 240
        try:
            while not ctx.join(timeout=5):
                if time.time() > deadline:
                    for p in ctx.processes:
                        p.kill()
                    pytest.fail('eight data-parallel workers did not finish within 240 s')
            break
        except ProcessExitedException as e:
            # Eight processes time-sharing ONE device is outside what this test is about (and what the runtime is tuned for): about one run
            # in four a worker is killed by the driver (HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION while its code objects load).  A worker that
            # dies by a SIGNAL is retried; a failed assertion inside a worker arrives as ProcessRaisedException and fails the test at once.
            if attempt == 2:
                raise
            print('retrying after a worker was killed by a signal: %s' % e)
    got = torch.load(out)
    worst = 0.0
    for k, g in ref.items():
        scale = float(g.abs().max())
        if scale < 1e-6:
            assert maxdiff(got['grads'][k], g) < 1e-6, k
            continue
        e = maxdiff(got['grads'][k], g) / scale
        worst = max(worst, e)
        assert e < 1e-4, (k, e)
    for k, v in ref_stats.items():
        assert maxdiff(got['stats'][k], v) < 1e-5, k
    print('worst relative gradient difference, 8 ranks vs single process: %.2e' % worst)


@pytest.mark.parametrize('N', [80, 128, 7])
def test_sync_bn_record_merge_kernels_equal_the_statistics_of_the_concatenated_batch(N):
    """the two launches of the SyncBN forward (local record; merge of the gathered records) on three ragged shards = batch
    statistics of all rows together, nn.BatchNorm1d's running update with the unbiased variance over the GLOBAL count, and the
    backward's division by that count read from the device"""
    from semi_tts_amd import ops
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(N)
    shards = [(torch.randn(m, N, generator=g) * 2 + 0.5).to(dev) for m in (37, 1, 300)]
    allrows = torch.cat(shards)
    tracked = torch.zeros(1, dtype=torch.int64, device=dev)
    recs = torch.stack([ops.bn_stats_record(x, 0, N, tracked if i == 0 else None) for i, x in enumerate(shards)])
    assert int(tracked) == 1
    assert [float(r[2 * N]) for r in recs] == [37.0, 1.0, 300.0]
    rm, rv = torch.full((N,), 0.25, device=dev), torch.full((N,), 2.0, device=dev)
    mean, var, inv_total = ops.bn_sync_merge(recs, N, rm, rv, 0.1)
    assert float(inv_total) == pytest.approx(1.0 / 338.0, rel=1e-7)
    assert maxdiff(mean, allrows.mean(0)) < 2e-6
    assert maxdiff(var, allrows.var(0, unbiased=False)) < 1e-5
    assert maxdiff(rm, 0.9 * 0.25 + 0.1 * allrows.mean(0)) < 1e-6
    assert maxdiff(rv, 0.9 * 2.0 + 0.1 * allrows.var(0, unbiased=True)) < 1e-5
    # one rank's record merged alone is that rank's own statistics
    m1, v1, it1 = ops.bn_sync_merge(recs[2:3].contiguous(), N)
    m0, v0 = ops.bn_stats(shards[2], 0, N)
    assert maxdiff(m1, m0) < 1e-6 and maxdiff(v1, v0) < 1e-6 and float(it1) == pytest.approx(1.0 / 300.0, rel=1e-7)
    # backward: the device-side 1 / count form is the host-side Mstat form
    x = shards[0]
    dy, w = torch.randn(37, N, generator=g).to(dev), torch.rand(N, generator=g).to(dev) + 0.5
    s = torch.randn(2 * N, generator=g).to(dev)
    a = ops.bn_bwd_apply(dy, None, None, x, mean, var, w, 1e-5, s, 338)
    b = ops.bn_bwd_apply(dy, None, None, x, mean, var, w, 1e-5, s, 37, inv_total)
    assert maxdiff(a, b) < 1e-6
