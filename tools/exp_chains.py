#!/usr/bin/env python3
"""Experiment (development tool): decode B=32 as N independent sub-batch chains, each its own hipGraph on its own
stream, launched concurrently.  The chains do not depend on each other, so there is no cross-stream event inside the loop;
the question is whether the latency-bound launches of one chain overlap with the other's."""
import os
import sys
import time
import ctypes as C

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch  # noqa: E402
from helpers import full_tacotron  # noqa: E402
from semi_tts_amd import ops, _lib  # noqa: E402
from semi_tts_amd.runtime import GraphedDecoder  # noqa: E402

dev = torch.device('cuda:0')
m = full_tacotron(dev, seed=7, prenet_dropout=0.5)
dec = m.decoder
lib = _lib.load()
B, L, frames = 32, 43, 258
mem = torch.randn(B, L, 512, device=dev) * 0.3
spk = torch.randn(B, 128, device=dev) * 0.3
for n in (1, 2, 4):
    b = B // n
    gds, streams = [], []
    for i in range(n):
        gd = GraphedDecoder(dec, b, L, frames, dev).capture()
        gd.memory.copy_(mem[i * b:(i + 1) * b]); gd.spkr.copy_(spk[i * b:(i + 1) * b])
        gds.append(gd)
        s = C.c_void_p(); _lib.check(lib.st_stream_create(C.byref(s)), 'stream'); streams.append(s)
    def run():
        for gd, s in zip(gds, streams):
            gd.draw_masks()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for gd, s in zip(gds, streams):
            with ops.use_stream(s.value):
                gd.graph.launch()
        for s in streams:
            lib.st_stream_sync(s)
        return time.perf_counter() - t0
    for _ in range(3):
        run()
    ts = [run() for _ in range(10)]
    t = sorted(ts)[len(ts) // 2]
    print('%d chain(s) of B=%2d: %.3f ms per 32x258 frames -> %.0f frames/s' % (n, b, t * 1e3, B * frames / t))
