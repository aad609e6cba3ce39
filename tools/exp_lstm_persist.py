#!/usr/bin/env python3
"""us per bidirectional LSTM layer: one launch for all steps (st_lstm_seq2_persist_fwd) against one launch per step, graph replays.
   python tools/exp_lstm_persist.py [B T H [train]] ..."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semi_tts_amd import ops

dev = torch.device('cuda')
shapes = [(32, 43, 256, 0), (32, 43, 256, 1), (64, 171, 256, 0), (32, 129, 256, 1)]
for B, T, H, train in shapes:
    xp = [torch.randn(B, T, 4 * H, device=dev) for _ in range(2)]
    w = [torch.randn(4 * H, H, device=dev) / H ** 0.5 for _ in range(2)]
    b = [torch.randn(4 * H, device=dev) * 0.1 for _ in range(2)]
    out = torch.zeros(B, T, 2 * H, device=dev)
    gs = [torch.zeros(T, B, 4, H, device=dev) for _ in range(2)] if train else None
    cs = [torch.zeros(T, B, H, device=dev) for _ in range(2)] if train else None
    res = {}
    for persist in (True, False):
        ops.LSTM_PERSIST = persist
        g = ops.Graph()
        with g.memory():
            ops.lstm_seq2(xp[0], xp[1], w[0], w[1], b[0], b[1], out, gs, cs)
            with g.capture():
                for _ in range(5):
                    ops.lstm_seq2(xp[0], xp[1], w[0], w[1], b[0], b[1], out, gs, cs)
        g.launch(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g.launch()
        torch.cuda.synchronize()
        res[persist] = (time.perf_counter() - t0) / 50 * 1e6
    ops.check_persist_status(dev)
    print('B=%d T=%d H=%d %s: one launch %.1f us (%.2f us per step) | per-step launches %.1f us (%.2f us per step)' % (
        B, T, H, 'training (tapes)' if train else 'inference', res[True], res[True] / T, res[False], res[False] / T), flush=True)
