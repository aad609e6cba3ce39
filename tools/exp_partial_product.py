#!/usr/bin/env python3
"""Timing of the K-split partial product (st_skinny_partial_attn_bwd without an attention job) against the plain packed product for the two
BPTT shapes (N = 2560 / 1792, K = 4096, B = 32), back to back on one stream with a 64 MB streaming write in between (cold operands)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semi_tts_amd import ops, _lib
from semi_tts_amd._lib import StT16View
lib = _lib.load()
dev = torch.device('cuda:0')
B, K = 32, 4096
flush = torch.empty(64 << 20, device=dev)
for N in (2560, 1792):
    w = torch.randn(N, K, device=dev) * 0.01
    pw = ops.pack_weight([w], [K], N)
    x = torch.randn(B, K, device=dev)
    xt = ops.tile_rows(x)
    xv = StT16View(ops._p(xt), K // 16, 0)
    y = torch.empty(B, N, device=dev)
    ref = x @ w.t()
    for S in (1, 2, 4, 8):
        part = torch.empty(S, B, N, device=dev)
        def run_part():
            _lib.check(lib.st_skinny_partial_attn_bwd(ops._p(pw), C.byref(xv), K, ops._p(part), S, B, N, None, ops.stream_handle()), 'part')
        def run_plain():
            _lib.check(lib.st_skinny_linear_packed_fwd(ops._p(pw), C.byref(xv), K, None, 0, None, 0, ops._p(y), N, None, 0, None, 0, 0, 0, 0, None, 0, None, B, N,
                                                       ops.stream_handle()), 'plain')
        for name, fn in (('partial S=%d' % S, run_part),) + ((('plain', run_plain),) if S == 1 else ()):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(20):
                flush.fill_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            err = float((part.sum(0) - ref).abs().max()) if name.startswith('partial') else float((y - ref).abs().max())
            print('N=%d %-14s median %.2f us (min %.2f)  max|err| %.2e' % (N, name, ts[len(ts) // 2], ts[0], err))
