#!/usr/bin/env python3
"""BiGRU (B=32, T=258, H=80) with its inputs cold (a 512 MB fill between launches evicts L2 / MALL) vs warm (back-to-back):
the in-situ launch of the whole forward (165 us) against the isolated one (130 us)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semi_tts_amd import ops, _lib
dev = torch.device('cuda')
B, T, H = 32, 258, 80
gi = [torch.randn(B, T, 3 * H, device=dev) for _ in range(2)]
w = [torch.randn(3 * H, H, device=dev) / H ** 0.5 for _ in range(2)]
b = [torch.randn(3 * H, device=dev) * 0.1 for _ in range(2)]
out = torch.zeros(B, T, 2 * H, device=dev)
big = torch.empty(128 * 1024 * 1024, device=dev)
lib = _lib.load()
def run():
    _lib.check(lib.st_gru_seq_fwd(ops._p(gi[0]), ops._p(gi[1]), ops._p(w[0]), ops._p(w[1]), ops._p(b[0]), ops._p(b[1]), ops._p(out), 2 * H,
                                  None, B, T, H, 2, ops.stream_handle()), 'gru')
for mode in ('warm', 'cold', 'rewritten'):
    ts = []
    for it in range(12):
        if mode == 'cold':
            big.fill_(1.0)
        elif mode == 'rewritten':          # the producer just wrote the inputs (as the in-projection GEMM does in the forward)
            gi[0].mul_(1.0); gi[1].mul_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    print('BiGRU %s inputs: median %.1f us (min %.1f, max %.1f)' % (mode, ts[len(ts) // 2], ts[0], ts[-1]))
