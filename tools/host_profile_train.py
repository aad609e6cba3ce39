#!/usr/bin/env python3
"""Host-side profile of the training step (cProfile over N steps after warm-up, statistics unread: the host never waits for the GPU).
usage: python tools/host_profile_train.py [--dist] [--steps N] > gpurun_out/host_profile.txt
Where the host's time per step goes once the GPU side is short enough for the host to be the bound (round 5)."""
import cProfile
import io
import os
import pstats
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault('ST_STATS_WINDOW_PERSIST', '1000')


def main():
    import argparse
    import torch
    import yaml
    from argparse import Namespace
    ap = argparse.ArgumentParser()
    ap.add_argument('--dist', action='store_true')
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--cycle', action='store_true', help='the VqvaeTrainer cycles (config 3) instead of the paired TTS step')
    ap.add_argument('--batch-size', type=int, default=None)
    args = ap.parse_args()
    import bench
    if args.dist:
        os.environ['ST_BENCH_FORCE_DIST'] = '1'
    rk = bench.Ranks(Namespace(gpus=1, dist=args.dist))
    from semi_tts_amd.solver import TtsTrainer, VqvaeTrainer
    config = yaml.safe_load(open(os.path.join(REPO, 'config', 'semi-single-spkr-paired-data.yaml' if args.cycle else 'semi-multi-spkr-paired-data.yaml')))
    paras = Namespace(batch_size=args.batch_size or bench.B, frames=bench.T_RAW, n_batches=1, seed=0, verbose=False, max_step=10 ** 9, load=None, n_spkr=109)
    tr = (VqvaeTrainer if args.cycle else TtsTrainer)(config, paras, 'train').load_data().set_model()
    tr.async_stats = True
    if args.cycle:
        pair, unpair = tr.fetch_data('pair_iter'), tr.fetch_data('unpair_iter')
        tr.step = 2

        def one():
            tr.cycle_step(pair, unpair if tr.cycle_kind(tr.step)[1] else None)
        tr.train_step = lambda *a: one()
        batch = ()
    else:
        batch = [t.to(rk.dev) for t in tr.batches[0]]
    for _ in range(8):
        tr.train_step(*batch)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr_bwd = cProfile.Profile()        # the backward functions run on the autograd engine's device thread: a profile of its own, switched on
    orig_backward = torch.Tensor.backward      # by a hook on the loss and off by an engine callback

    def backward(self, *a, **k):
        def on(_g):
            pr_bwd.enable()
            torch.autograd.Variable._execution_engine.queue_callback(pr_bwd.disable)
        self.register_hook(on)
        return orig_backward(self, *a, **k)
    torch.Tensor.backward = backward
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(args.steps):
        tr.train_step(*batch)
    pr.disable()
    torch.Tensor.backward = orig_backward
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('host issue %.3f ms per step (under cProfile), GPU finished %.3f ms after the last issue' % ((t1 - t0) / args.steps * 1e3, (t2 - t1) * 1e3))
    for key in ('tottime', 'cumtime'):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
        print(s.getvalue())
    print('==== the autograd thread (backward functions)')
    s = io.StringIO()
    pstats.Stats(pr_bwd, stream=s).sort_stats('tottime').print_stats(40)
    print(s.getvalue())
    s = io.StringIO()
    pstats.Stats(pr_bwd, stream=s).sort_stats('cumtime').print_stats(40)
    print(s.getvalue())
    rk.close()


if __name__ == '__main__':
    main()
