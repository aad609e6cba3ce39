#!/usr/bin/env python3
"""Stress of the in-launch hand-off (query projection -> attention fin part): thousands of graph replays and eager passes of the
C2 decode loop must stay finite and bit-identical (a spin time-out would poison the outputs with NaN)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from helpers import full_tacotron
from semi_tts_amd.runtime import GraphedDecoder
from semi_tts_amd.synthetic import synthetic_batch

dev = torch.device('cuda:0')
n_replays = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for B, L, T in ((32, 43, 258), (64, 100, 99), (7, 12, 60)):
    m = full_tacotron(dev, seed=11, prenet_dropout=0.5)
    txt, spk, _ = synthetic_batch(B, L, T, seed=5)
    txt, spk = torch.from_numpy(txt).to(dev), torch.from_numpy(spk).to(dev)
    with torch.no_grad():
        mem = m.encoder(txt, None).contiguous()
    gd = GraphedDecoder(m.decoder, B, L, T, dev).capture()
    ref = [t.clone() for t in gd(mem, spk, redraw=True)]
    torch.cuda.synchronize()
    t0 = time.time()
    bad = 0
    for i in range(n_replays):
        out = gd(redraw=False)
        if i % 50 == 49 or i == n_replays - 1:
            torch.cuda.synchronize()
            bad += int(not all(torch.equal(a, b) for a, b in zip(out, ref)))
    torch.cuda.synchronize()
    with torch.no_grad():
        for i in range(20):
            e = m.decoder(mem, None, T, spk, _masks={'own': gd.own_mask})
            bad += int(not all(torch.equal(a, b) for a, b in zip(e, ref)))
    print('B=%d L=%d steps=%d: %d replays + 20 eager passes in %.1f s, finite=%s, mismatching checks=%d' %
          (B, L, T // 3, n_replays, time.time() - t0, bool(torch.isfinite(ref[0]).all()), bad))
    assert bad == 0 and bool(torch.isfinite(ref[0]).all())
print('ok')
