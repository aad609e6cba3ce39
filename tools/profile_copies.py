#!/usr/bin/env python3
"""Development tool (MI355X box): torch.profiler over one training step -- which ATen ops launch the copy / fill kernels, and
from where (Python stack or autograd node)."""
import collections
import os
import sys
from argparse import Namespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import yaml
from torch.profiler import profile, ProfilerActivity
from semi_tts_amd.solver import TtsTrainer

config = yaml.safe_load(open(os.path.join(ROOT, 'config', 'semi-multi-spkr-paired-data.yaml')))
paras = Namespace(batch_size=32, frames=256, n_batches=1, seed=0, verbose=False, max_step=10 ** 9, load=None, n_spkr=109)
tr = TtsTrainer(config, paras, 'train').load_data().set_model()
batch = [t.to(tr.device) for t in tr.batches[0]]
tr.train_step(*batch)
tr.train_step(*batch)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(*batch)
    torch.cuda.synchronize()
by_op = collections.Counter()
by_site = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::clone', 'aten::fill_', 'aten::zero_', 'aten::contiguous', 'aten::add_', 'aten::cat', 'aten::mul', 'aten::add'):
        by_op[ev.name] += 1
        st = [s for s in (ev.stack or []) if 'semi_tts_amd' in s or 'solver' in s or 'optim' in s]
        by_site[(ev.name, st[0].split('/')[-1] if st else ('<autograd/other> ' + (ev.stack[0].split('/')[-1] if ev.stack else '')))] += 1
print('ops:', dict(by_op))
for k, n in sorted(by_site.items(), key=lambda kv: -kv[1])[:50]:
    print('%5d  %-16s %s' % (n, k[0], k[1]))
kern = collections.Counter()
for ev in prof.events():
    if ev.device_type is not None and 'cuda' in str(ev.device_type).lower():
        kern[ev.name[:90]] += 1
print('--- kernels by launch count')
for k, n in kern.most_common(25):
    print('%5d  %s' % (n, k))
