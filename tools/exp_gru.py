#!/usr/bin/env python3
"""us per CBHG BiGRU sequence (B=32, T=258, H=80), inference and training form, graph replays"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semi_tts_amd import ops, _lib
import ctypes as C
dev = torch.device('cuda')
B, T, H = 32, 258, 80
gi = [torch.randn(B, T, 3 * H, device=dev) for _ in range(2)]
w = [torch.randn(3 * H, H, device=dev) / H ** 0.5 for _ in range(2)]
b = [torch.randn(3 * H, device=dev) * 0.1 for _ in range(2)]
out = torch.zeros(B, T, 2 * H, device=dev)
lib = _lib.load()
for tape in (None, torch.zeros(2, B, T, 4, H, device=dev)):
    def run():
        _lib.check(lib.st_gru_seq_fwd(ops._p(gi[0]), ops._p(gi[1]), ops._p(w[0]), ops._p(w[1]), ops._p(b[0]), ops._p(b[1]), ops._p(out), 2 * H,
                                      ops._p(tape), B, T, H, 2, ops.stream_handle()), 'gru')
    run()
    g = ops.Graph()
    with g.capture():
        for _ in range(5):
            run()
    g.launch(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.launch()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 50 * 1e6
    print('BiGRU B=%d T=%d H=%d %s: %.1f us per sequence, %.3f us per step; checksum %.6f' % (B, T, H, 'training' if tape is not None else 'inference', us, us / T, float(out.sum())))
