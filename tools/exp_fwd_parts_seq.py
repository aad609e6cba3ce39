#!/usr/bin/env python3
"""Development tool (MI355X box, under rocprofv3 --kernel-trace): one eager encoder pass and one eager postnet pass at C2, bracketed
by marker launches (ops.fill_ of 1 / 2 / 3 floats) so that tools/prof_seq.py can cut them out of the trace."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from helpers import full_tacotron
from semi_tts_amd.synthetic import synthetic_batch

dev = torch.device('cuda')
B, L, T = 32, 43, 258
m = full_tacotron(dev, seed=1234, prenet_dropout=0.5)
txt, spk, _ = synthetic_batch(B, L, T, seed=100)
txt, spk = torch.from_numpy(txt).to(dev), torch.from_numpy(spk).to(dev)
mel = torch.randn(B, T, 80, device=dev)
with torch.no_grad():
    for _ in range(2):
        mem = m.encoder(txt, None)
        post = m.postnet(mel)
    torch.cuda.synchronize()
    torch.cuda._sleep(10)
    print('MARK encoder')
    import time
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mem = m.encoder(txt, None)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        post = m.postnet(mel)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print('encoder %.1f us, postnet %.1f us' % (1e6 * (t1 - t0), 1e6 * (t2 - t1)))
