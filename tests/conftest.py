import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    """-> (weights dict of torch tensors, arrays dict (lists for 'key/NNN' groups), meta dict)"""
    import torch
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    weights, arrays, groups = {}, {}, {}
    for k in z.files:
        if k == 'meta':
            continue
        v = z[k]
        if k.startswith('w/'):
            weights[k[2:]] = torch.from_numpy(v)
        elif '/' in k:
            g, i = k.rsplit('/', 1)
            if i.isdigit():
                groups.setdefault(g, {})[int(i)] = torch.from_numpy(v)
            else:                                  # a named group ('ckpt_asr/<state-dict key>'): dict of tensors
                arrays.setdefault(g, {})[i] = torch.from_numpy(v)
        else:
            arrays[k] = torch.from_numpy(v) if v.dtype != np.uint8 else v
    for g, d in groups.items():
        arrays[g] = [d[i] for i in sorted(d)]
    meta = json.loads(bytes(z['meta']).decode()) if 'meta' in z.files else {}
    return weights, arrays, meta


@pytest.fixture(scope='session')
def golden():
    return load_golden
