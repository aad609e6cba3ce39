#!/usr/bin/env python3
"""Microbenchmark of ops.gemm (fp32 MFMA GEMM / implicit-GEMM conv) on the shapes of one C2 forward: us per call and TFLOP/s
against the 157.3 TFLOP/s fp32 matrix peak."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semi_tts_amd import ops

dev = torch.device('cuda')
SHAPES = [  # name, Bn, T, Cin, N, KT, pad
    ('enc conv k5 512->512', 32, 43, 512, 512, 5, 2), ('enc conv k5 64->512', 32, 43, 64, 512, 5, 2),
    ('enc lstm in-proj 512->1024', 32, 43, 512, 1024, 1, 0), ('memory layer 512->256', 32, 43, 512, 256, 1, 0),
    ('bank conv k1 80->80', 32, 258, 80, 80, 1, 0), ('bank conv k4 80->80', 32, 258, 80, 80, 4, 2), ('bank conv k8 80->80', 32, 258, 80, 80, 8, 4),
    ('proj conv k3 640->128', 32, 258, 640, 128, 3, 1), ('proj conv k3 128->80', 32, 258, 128, 80, 3, 1),
    ('highway 80->80', 32, 258, 80, 80, 1, 0), ('gru in-proj 80->240', 32, 258, 80, 240, 1, 0), ('linear 160->1025', 32, 258, 160, 1025, 1, 0),
    ('teacher prenet 240->256', 32, 86, 240, 256, 1, 0)]
rows = []
for name, Bn, T, Cin, N, KT, pad in SHAPES:
    x = torch.randn(Bn, T, Cin, device=dev)
    w = torch.randn(N, Cin, KT, device=dev) if KT > 1 else torch.randn(N, Cin, device=dev)
    out = ops.gemm(x, w, pad=pad)
    g = ops.Graph()
    with g.capture():
        for _ in range(20):
            ops.gemm(x, w, out, pad=pad)
    g.launch(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.launch()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 200 * 1e6
    M = Bn * out.shape[1]
    flop = 2.0 * M * N * Cin * KT
    rows.append(dict(shape=name, M=M, N=N, K=Cin * KT, us=round(us, 2), tflops=round(flop / us / 1e6, 1), frac=round(flop / us / 1e6 / 157.3, 3),
                     workgroups=((M + 63) // 64) * ((N + 63) // 64)))
    print(rows[-1])
print(json.dumps(rows))
