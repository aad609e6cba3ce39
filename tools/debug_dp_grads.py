#!/usr/bin/env python3
"""Which gradients / weights differ between the plain trainer and the trainer with the GradReducer attached (world size 1, gloo or
nccl, collectives forced), step by step.  usage: python tools/debug_dp_grads.py [--steps 3] [--backend nccl]"""
import argparse, os, socket, sys
from argparse import Namespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--backend', default='nccl')
    a = ap.parse_args()
    import torch, yaml
    import torch.distributed as dist
    from semi_tts_amd import parallel
    from semi_tts_amd.solver import TtsTrainer
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', str(port))
    dist.init_process_group(a.backend, rank=0, world_size=1, **({'device_id': dev} if a.backend == 'nccl' else {}))
    config = yaml.safe_load(open(os.path.join(ROOT, 'config', 'semi-multi-spkr-paired-data.yaml')))

    def run(force):
        parallel.force_collectives(force)
        paras = Namespace(batch_size=32, frames=256, n_batches=1, seed=0, verbose=False, max_step=10 ** 9, load=None, n_spkr=109)
        tr = TtsTrainer(config, paras, 'train').load_data().set_model()
        parallel.sync_batchnorm(False)
        batch = [t.to(dev) for t in tr.batches[0]]
        torch.manual_seed(77)
        out = []
        orig = tr.clip_grad_norm_
        snap = {}

        def spy(params, max_norm, **kw):
            snap['g'] = {k: p.grad.detach().clone() for k, p in tr.model.named_parameters() if p.grad is not None}
            return orig(params, max_norm, **kw)
        tr.clip_grad_norm_ = spy
        for i in range(a.steps):
            st = tr.train_step(*batch)
            torch.cuda.synchronize()
            out.append(dict(st=dict(st), grads=snap['g'], clipped={k: p.grad.detach().clone() for k, p in tr.model.named_parameters() if p.grad is not None},
                            weights={k: p.detach().clone() for k, p in tr.model.named_parameters()},
                            red=dict(tr.reducer.stats, sparse=tr.reducer._sparse) if tr.reducer is not None else None))
        if tr.reducer is not None:
            tr.reducer.close()
        return out
    plain, red = run(False), run(True)
    for i in range(a.steps):
        p, r = plain[i], red[i]
        print('step', i, 'loss', p['st']['loss'], r['st']['loss'], 'gn', p['st']['grad_norm'], r['st']['grad_norm'], r['red'])
        for what in ('grads', 'clipped', 'weights'):
            bad = [(k, float((p[what][k] - r[what][k]).abs().max())) for k in p[what] if k in r[what] and not torch.equal(p[what][k], r[what][k])]
            missing = set(p[what]) ^ set(r[what])
            print('  ', what, 'differ:', len(bad), bad[:12], 'missing:', sorted(missing)[:5])
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
