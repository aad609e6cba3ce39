#!/usr/bin/env python3
"""First execution of the RCCL branch on a ONE-GPU box: a world-size-1 `nccl` process group with every collective forced
(parallel.force_collectives), driving the C4 per-rank workload -- TtsTrainer.train_step at C2 size, 109 speakers,
config/semi-multi-spkr-paired-data.yaml -- through the hook-driven GradReducer (asynchronous all-reduce on RCCL's stream),
SyncBN's all_gather_into_tensor / all-reduce and the timing helper.  Sums over one rank are the identity, so:

  * reducer only (SyncBN off): loss, grad norm, every gradient and every updated weight are BITWISE those of the plain trainer
    after three steps (gradients are born in / gathered into their all-reduce buckets; nothing is zeroed or accumulated);
  * reducer + SyncBN: the merged statistics are (mean * M) / M etc., equal to one rounding (<= 1e-6), the loss to 1e-6.

Run as its own process (tests/test_gpu_rccl.py starts it as a child so that an RCCL failure cannot take pytest down).
Prints one JSON line; exit code 0 only if every check holds.   ref: BaseSolver.backward src/solver.py:138-151
"""
import argparse
import json
import os
import socket
import sys
import time
from argparse import Namespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch-size', type=int, default=32)
    ap.add_argument('--frames', type=int, default=256)
    ap.add_argument('--backend', default='nccl')
    ap.add_argument('--cycle', action='store_true', help='the VqvaeTrainer cycles (config 3 / 4 as the reference trains them) instead of the paired TTS step')
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    import yaml
    from semi_tts_amd import parallel
    from semi_tts_amd.solver import TtsTrainer
    assert torch.cuda.is_available(), 'needs a GPU'
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(free_port()))
    t0 = time.perf_counter()
    kw = {'device_id': dev} if a.backend == 'nccl' else {}
    dist.init_process_group(a.backend, rank=0, world_size=1, **kw)
    t_init = time.perf_counter() - t0
    config = yaml.safe_load(open(os.path.join(ROOT, 'config', 'semi-multi-spkr-paired-data.yaml')))
    if a.cycle:
        return cycle_main(a, config, dev)

    def run(force, syncbn):
        parallel.force_collectives(force)
        paras = Namespace(batch_size=a.batch_size, frames=a.frames, n_batches=1, seed=0, verbose=False, max_step=10 ** 9,
                          load=None, n_spkr=109)
        tr = TtsTrainer(config, paras, 'train').load_data().set_model()
        parallel.sync_batchnorm(syncbn)
        batch = [t.to(dev) for t in tr.batches[0]]
        torch.manual_seed(77)                                     # the same dropout masks in every variant
        st = tr.train_step(*batch)
        torch.cuda.synchronize()
        counts = parallel.collective_counts()
        counts['async_grad_buckets'] = parallel.async_bucket_count()
        first = dict(st)
        stats1 = {k: v.detach().clone() for k, v in tr.model.state_dict().items() if 'running_' in k}
        grads1 = {k: p.grad.detach().clone() for k, p in tr.model.named_parameters() if p.grad is not None}
        # two more steps: the reducer's second form (static graph: buckets in gradient-arrival order, one hook per bucket) must
        # reproduce the plain trainer's third step just as well
        for _ in range(2):
            st = tr.train_step(*batch)
        torch.cuda.synchronize()
        later = dict(counts=parallel.collective_counts(), reducer=dict(tr.reducer.stats, sparse=tr.reducer._sparse,
                     buckets=len(tr.reducer.buckets), dropped=len(tr.reducer.dropped)) if tr.reducer is not None else None)
        grads = {k: p.grad.detach().clone() for k, p in tr.model.named_parameters() if p.grad is not None}
        weights = {k: p.detach().clone() for k, p in tr.model.named_parameters()}
        stats = {k: v.detach().clone() for k, v in tr.model.state_dict().items() if 'running_' in k}
        had_reducer = tr.reducer is not None
        if tr.reducer is not None:
            tr.reducer.close()
        return dict(st=first, st3=st, counts=counts, grads=grads, weights=weights, stats=stats, reducer=had_reducer, later=later, stats1=stats1, grads1=grads1)

    plain = run(False, False)
    red = run(True, False)
    full = run(True, True)
    # max over ranks of a device scalar through the group (what bench.py's timing contract does)
    parallel.force_collectives(True)
    t_max = parallel.max_over_ranks(1.25, device=dev)
    dist.barrier()

    def bitwise(x, y):
        return set(x) == set(y) and all(torch.equal(x[k], y[k]) for k in x)

    def worst(x, y):
        return max(float((x[k] - y[k]).abs().max()) / max(float(y[k].abs().max()), 1e-12) for k in y)

    res = {
        'rccl_ranks': 1 if a.backend == 'nccl' else 0, 'backend': dist.get_backend(), 'init_seconds': round(t_init, 2),
        'workload': 'TtsTrainer.train_step, B=%d, %d frames, 109 speakers (C4 per-rank workload)' % (a.batch_size, a.frames),
        'plain': {k: plain['st'][k] for k in ('loss', 'grad_norm')},
        'reducer_only': {'loss': red['st']['loss'], 'grad_norm': red['st']['grad_norm'], 'collectives_per_step': red['counts'],
                         'reducer_attached': red['reducer'], 'third_step': red['later'],
                         'third_step_loss_equal': red['st3']['loss'] == plain['st3']['loss'] and red['st3']['grad_norm'] == plain['st3']['grad_norm'],
                         'gradients_bitwise_equal': bitwise(red['grads'], plain['grads']),
                         'updated_weights_bitwise_equal': bitwise(red['weights'], plain['weights'])},
        'reducer_syncbn': {'loss': full['st']['loss'], 'grad_norm': full['st']['grad_norm'], 'collectives_per_step': full['counts'],
                           'loss_abs_diff': abs(full['st']['loss'] - plain['st']['loss']),
                           'grad_norm_rel_diff': abs(full['st']['grad_norm'] - plain['st']['grad_norm']) / plain['st']['grad_norm'],
                           'running_stats_max_rel_diff': worst(full['stats1'], plain['stats1']),
                           'gradient_max_rel_diff': worst(full['grads1'], plain['grads1'])},
        'max_over_ranks': t_max,
    }
    ok = (res['reducer_only']['reducer_attached'] and res['reducer_only']['gradients_bitwise_equal']
          and res['reducer_only']['updated_weights_bitwise_equal']
          and red['st']['loss'] == plain['st']['loss'] and red['st']['grad_norm'] == plain['st']['grad_norm']
          and res['reducer_only']['third_step_loss_equal'] and red['later']['reducer']['sparse'] and red['later']['reducer']['zeroed'] == 0
          and res['reducer_only']['collectives_per_step']['grad_buckets'] == 4
          and res['reducer_syncbn']['collectives_per_step'] == {'grad_buckets': 4, 'syncbn_fwd': 6, 'syncbn_bwd': 6,
                                                                'async_grad_buckets': 4 if a.backend == 'nccl' else 0}
          and res['reducer_syncbn']['loss_abs_diff'] <= 1e-6 * max(1.0, abs(plain['st']['loss']))
          and res['reducer_syncbn']['running_stats_max_rel_diff'] <= 1e-6
          and res['reducer_syncbn']['grad_norm_rel_diff'] <= 1e-5
          and t_max == 1.25)
    res['ok'] = bool(ok)
    print(json.dumps(res))
    sys.stdout.flush()
    dist.destroy_process_group()
    return 0 if ok else 1


def cycle_main(a, config, dev):
    """config 4 as the reference trains it (main.py:61-63: VqvaeTrainer on the multi-speaker configuration): four alternating cycle steps --
    speech-first with the unpaired batch, text-first -- through the dynamic GradReducer (the cycles' graphs depend on the data) with the
    all-reduces on RCCL at world size 1: losses, gradient norms, every gradient and every weight bitwise those of the plain trainer; with
    SyncBN on top (speech encoder's six BatchNorms included) to one rounding."""
    import torch
    import torch.distributed as dist
    from semi_tts_amd import parallel
    from semi_tts_amd.solver import VqvaeTrainer

    def run(force, syncbn):
        parallel.force_collectives(force)
        paras = Namespace(batch_size=a.batch_size, frames=a.frames, n_batches=1, seed=0, verbose=False, max_step=10 ** 9, load=None, n_spkr=109)
        tr = VqvaeTrainer(config, paras, 'train').load_data().set_model()
        parallel.sync_batchnorm(syncbn)
        pair, unpair = tr.fetch_data('pair_iter'), tr.fetch_data('unpair_iter')
        torch.manual_seed(77)
        tr.step = 2
        sts = []
        for _ in range(4):
            st = tr.cycle_step(pair, unpair if tr.cycle_kind(tr.step)[1] else None)
            sts.append({k: float(st[k]) for k in ('loss', 'grad_norm', 'asr_loss', 'tts_loss')})
        torch.cuda.synchronize()
        counts = parallel.collective_counts()
        out = dict(sts=sts, counts=counts, reducer=tr.reducer is not None,
                   grads={k: p.grad.detach().clone() for k, p in tr.model.named_parameters() if p.grad is not None},
                   weights={k: p.detach().clone() for k, p in tr.model.named_parameters()})
        if tr.reducer is not None:
            out['stats'] = dict(tr.reducer.stats)
            tr.reducer.close()
        return out

    plain, red, full = run(False, False), run(True, False), run(True, True)
    bitwise = lambda x, y: set(x) == set(y) and all(torch.equal(x[k], y[k]) for k in x)
    rel = max(abs(f['loss'] - p['loss']) / max(1.0, abs(p['loss'])) for f, p in zip(full['sts'], plain['sts']))
    res = {'rccl_ranks': 1 if a.backend == 'nccl' else 0, 'backend': dist.get_backend(),
           'workload': 'VqvaeTrainer.cycle_step x 4 (speech-first with the unpaired batch / text-first), B=%d + %d, %d frames, 109 speakers' % (a.batch_size, a.batch_size, a.frames),
           'plain': plain['sts'], 'reducer_attached': red['reducer'], 'reducer_stats_last_step': red.get('stats'),
           'statistics_equal': red['sts'] == plain['sts'],
           'gradients_bitwise_equal': bitwise(red['grads'], plain['grads']), 'updated_weights_bitwise_equal': bitwise(red['weights'], plain['weights']),
           'collectives_last_step': red['counts'], 'syncbn_collectives_last_step': full['counts'], 'syncbn_loss_max_rel_diff': rel}
    # (text-first, the last step: 6 speech-encoder + 3 text-encoder + 1 bank + 2 projection BatchNorm gathers)
    ok = (res['reducer_attached'] and res['statistics_equal'] and res['gradients_bitwise_equal'] and res['updated_weights_bitwise_equal']
          and full['counts']['syncbn_fwd'] == 12 and full['counts']['syncbn_bwd'] == 12 and rel <= 1e-5)
    res['ok'] = bool(ok)
    print(json.dumps(res))
    sys.stdout.flush()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
