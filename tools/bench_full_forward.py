#!/usr/bin/env python3
"""Secondary measurement: whole Tacotron2.forward (encoder + decoder + CBHG postnet + linear) at C2, free-running
inference: mel-frames/s of one hipGraph replay of the whole forward, of eager launches, and a split by part."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from helpers import full_tacotron
from semi_tts_amd.synthetic import synthetic_batch

dev = torch.device('cuda')
B, L, T = 32, 43, 258
m = full_tacotron(dev, seed=1234, prenet_dropout=0.5)
m.decoder.cache_packed = True
txt, spk, _ = synthetic_batch(B, L, T, seed=100)
txt, spk = torch.from_numpy(txt).to(dev), torch.from_numpy(spk).to(dev)
sync = torch.cuda.synchronize


def timed(fn, n=10):
    for _ in range(3):
        fn()
    sync(); t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    sync()
    return (time.perf_counter() - t0) / n, out


with torch.no_grad():
    t_all, out = timed(lambda: m(txt, None, T, spk, tf_rate=0.0))
    t_enc, mem = timed(lambda: m.encoder(txt, None))
    t_dec, dout = timed(lambda: m.decoder(mem, None, T, spk, tf_rate=0.0))
    t_post, _ = timed(lambda: m.postnet(dout[0]))
from semi_tts_amd.runtime import GraphedTacotron2
gt = GraphedTacotron2(m, B, L, T, dev)
gt.txt.copy_(txt); gt.spkr.copy_(spk)
gt.capture()
t_graph, gout = timed(lambda: gt(redraw=True))
print(json.dumps(dict(metric='mel-frames/sec (whole Tacotron2.forward, one hipGraph replay)', value=B * T / t_graph, ms_total_graph=1e3 * t_graph,
                      value_eager=B * T / t_all, ms_total=1e3 * t_all,
                      ms_encoder=1e3 * t_enc, ms_decoder_eager=1e3 * t_dec, ms_postnet=1e3 * t_post,
                      config='C2: B=32, 258 frames, L=43, fp32, prenet dropout 0.5')))
