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
from semi_tts_amd import ops as _ops               # no starved hand-off / one-launch layer in the timed replays
assert not _ops.handoff_starved(m.decoder.handoff_status) and not _ops.persist_starved(), 'a timed replay was starved of compute units' 
assert bool(torch.isfinite(gout[0]).all()) and bool(torch.isfinite(gout[1]).all())
# roofline of the whole forward: SURVEY 8d counts 132 GFLOP per C2 batch (encoder 12.4 + decoder 106.3 + postnet 13.7, dense
# contractions on the fp32 matrix cores) and, per decode step, 81.0 MB of operands that must stream (weights re-read every step)
import bench
flops = 12.4e9 + 106.3e9 + 13.7e9
hbm_bytes = 86 * 81.0e6 + 33.9e6
res = dict(metric='mel-frames/sec (whole Tacotron2.forward, one hipGraph replay)', value=B * T / t_graph, unit='mel-frames/s',
           ms_total_graph=1e3 * t_graph, value_eager=B * T / t_all, ms_total=1e3 * t_all,
           ms_encoder=1e3 * t_enc, ms_decoder_eager=1e3 * t_dec, ms_postnet=1e3 * t_post, dtype='f32', data='synthetic',
           config={'workload': 'C2 whole forward: encoder + decode loop + CBHG postnet + Linear(160, 1025), B=32, 258 frames, L=43, prenet dropout 0.5'},
           roofline={'bound': 'hbm', 'kernel': 'whole forward (the decode loop streams its weights every step)',
                     'achieved': round(hbm_bytes / t_graph / 1e9, 1), 'peak': bench.HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(hbm_bytes / t_graph / 1e9 / bench.HBM_PEAK_GBS, 4), 'traffic': None,
                     'mfma': {'achieved': round(flops / t_graph / 1e12, 2), 'peak': bench.MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                              'frac': round(flops / t_graph / 1e12 / bench.MFMA_F32_PEAK_TFLOPS, 4)}})
if '--no-cpu-baseline' not in sys.argv:
    from oracle import nn_baseline as NB
    from helpers import full_hp
    W = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    net = NB.NNTacotron2(W, full_hp(0.5)).eval()
    txt_c, spk_c = txt.cpu(), spk.cpu()
    res['cpu_baseline'] = bench.cpu_timed(lambda: net(txt_c[:4], 12, spk_c[:4]), lambda i: net(txt_c, T, spk_c, seed=i), B * T, 'mel-frames/s',
                                          'nn_modules', 'full passes of Tacotron2.forward (B=%d, %d frames, L=%d, free running, prenet dropout 0.5) '
                                          'assembled from torch.nn modules (oracle/nn_baseline.py); probe = 4 utterances x 4 decode steps' % (B, T, L))
print(json.dumps(res))
