#!/usr/bin/env python3
"""Edge-shape sweep of the HIP decode path (development tool): tiny / ragged / long inputs either run and agree with the
oracle or fail loudly with a RuntimeError from the C ABI -- never silently."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from helpers import full_tacotron, full_hp
from oracle import tts_oracle as O

dev = torch.device('cuda')
m = full_tacotron(dev, seed=5)
W = {k: v.detach().cpu() for k, v in m.state_dict().items()}
for B, L, T in [(1, 1, 3), (1, 5, 6), (2, 2, 9), (3, 200, 12), (17, 43, 30), (33, 7, 6), (5, 400, 6), (2, 900, 6), (2, 3000, 6)]:
    g = torch.Generator().manual_seed(B * 1000 + L)
    txt = torch.randn(B, L, 64, generator=g)
    spk = torch.randn(B, 128, generator=g)
    try:
        with torch.no_grad():
            mel, lin, align, stop = m(txt.to(dev), None, T, spk.to(dev), tf_rate=0.0)
        torch.cuda.synchronize()
    except RuntimeError as e:
        print('B=%d L=%d T=%d -> RuntimeError: %s' % (B, L, T, str(e)[:160]))
        continue
    with torch.no_grad():
        mel_r, lin_r, align_r, _ = O.tacotron2_forward(W, txt, T, spk, full_hp(0.0))
    print('B=%d L=%d T=%d -> ok mel %.2e lin %.2e align %.2e finite=%s' % (
        B, L, T, float((mel.cpu() - mel_r).abs().max()), float((lin.cpu() - lin_r).abs().max()),
        float((align.cpu() - align_r).abs().max()), bool(torch.isfinite(mel).all())))
