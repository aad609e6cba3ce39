#!/usr/bin/env python3
"""Development tool (MI355X box): which Python call sites issue the torch copy / fill / cat launches of one training step?
Patches the torch entry points that end in a device copy or fill and counts them by caller (file:line) over one TtsTrainer step."""
import collections
import os
import sys
import traceback
from argparse import Namespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import yaml

counts = collections.Counter()
active = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'semi_tts_amd' in fr.filename or fr.filename.endswith('bench.py'):
            return '%s:%d' % (os.path.basename(fr.filename), fr.lineno)
    fr = traceback.extract_stack()[-3]
    return '%s:%d' % (os.path.basename(fr.filename), fr.lineno)


def wrap(obj, name, label, pred=None):
    orig = getattr(obj, name)

    def f(*a, **k):
        if active[0] and (pred is None or pred(*a, **k)):
            counts[(label, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)


wrap(torch.Tensor, 'copy_', 'copy_')
wrap(torch.Tensor, 'clone', 'clone')
wrap(torch.Tensor, 'contiguous', 'contiguous(copy)', lambda t, *a, **k: not t.is_contiguous())
wrap(torch.Tensor, 'zero_', 'zero_')
wrap(torch.Tensor, 'fill_', 'fill_')
wrap(torch, 'zeros', 'zeros')
wrap(torch, 'zeros_like', 'zeros_like')
wrap(torch, 'cat', 'cat')
wrap(torch, 'stack', 'stack')
wrap(torch.Tensor, 'sum', 'sum')
wrap(torch.Tensor, '__add__', 'add')
wrap(torch.Tensor, '__mul__', 'mul')
wrap(torch.Tensor, '__iadd__', 'iadd')

from semi_tts_amd.solver import TtsTrainer
config = yaml.safe_load(open(os.path.join(ROOT, 'config', 'semi-multi-spkr-paired-data.yaml')))
paras = Namespace(batch_size=32, frames=256, n_batches=1, seed=0, verbose=False, max_step=10 ** 9, load=None, n_spkr=109)
tr = TtsTrainer(config, paras, 'train').load_data().set_model()
batch = [t.to(tr.device) for t in tr.batches[0]]
tr.train_step(*batch)
active[0] = True
tr.train_step(*batch)
active[0] = False
torch.cuda.synchronize()
tot = collections.Counter()
for (label, s), n in counts.items():
    tot[label] += n
print('totals:', dict(tot))
for (label, s), n in sorted(counts.items(), key=lambda kv: -kv[1])[:60]:
    print('%5d  %-18s %s' % (n, label, s))
