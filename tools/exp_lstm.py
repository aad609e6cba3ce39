#!/usr/bin/env python3
"""Timing experiment (MI355X box): the two LSTM-cell launches of a C2 decode step replayed from a hipGraph as (a) query cell only,
(b) decoder cell only, (c) alternating as in the loop.  (a) re-reads the same 29.8 MB every launch (3.7 MB per XCD: it fits the
4 MB L2s), so (a) against (c) says whether weights left in an XCD's L2 by one launch are still there for the next one -- the
premise of warming the L2 from idle compute units of the preceding launch.  Prints one line per pattern."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import bench
from helpers import full_tacotron
from semi_tts_amd import _lib, ops

dev = torch.device('cuda:0')
Bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dec = full_tacotron(dev, seed=1234, prenet_dropout=0.5).decoder
lib = _lib.load()
Q, D, E, P = dec.query_rnn_dim, dec.dec_rnn_dim, dec.enc_embed_dim, dec.prenet_dim
f32 = dict(device=dev, dtype=torch.float32)
wq_ih, wd_ih = dec.query_rnn.weight_ih, dec.dec_rnn.weight_ih
pk_q = ops.pack_weight([wq_ih, wq_ih[:, P:], dec.query_rnn.weight_hh], [P, E, Q], 4 * Q, lstm_H=Q, ldws=[P + E, P + E, Q])
pk_d = ops.pack_weight([wd_ih, wd_ih[:, E:], dec.dec_rnn.weight_hh], [E, Q, D], 4 * D, lstm_H=D, ldws=[E + Q, E + Q, D])
Kq, Kd = P + E + Q, E + Q + D
xq, xd = ops.tile_rows(torch.randn(Bsz, Kq, **f32)), ops.tile_rows(torch.randn(Bsz, Kd, **f32))
c_q, c_d = torch.randn(Bsz, Q, **f32), torch.randn(Bsz, D, **f32)
ho, co = torch.zeros(ops.t16_floats(Bsz, Q), **f32), torch.empty(Bsz, Q, **f32)
xq_v, xd_v, ho_v = ops.t16_view(xq, K=Kq), ops.t16_view(xd, K=Kd), ops.t16_view(ho, K=Q)


def q():
    ops.lstm_cell_packed(pk_q, xq_v, Kq, dec.query_rnn.bias_ih, dec.query_rnn.bias_hh, c_q, ho_v, co, Bsz, Q)


def d():
    ops.lstm_cell_packed(pk_d, xd_v, Kd, dec.dec_rnn.bias_ih, dec.dec_rnn.bias_hh, c_d, ho_v, co, Bsz, D)


for name, seq, nbytes in (('query cell only (29.8 MB)', (q,), bench.lstm_algorithmic_bytes(Bsz, Q, Kq)),
                          ('decoder cell only (42.9 MB)', (d,), bench.lstm_algorithmic_bytes(Bsz, D, Kd)),
                          ('alternating', (q, d), 0.5 * (bench.lstm_algorithmic_bytes(Bsz, Q, Kq) + bench.lstm_algorithmic_bytes(Bsz, D, Kd)))):
    g = ops.Graph()
    for f in seq:
        f()
    inner = 60
    with g.capture():
        for _ in range(inner // len(seq)):
            for f in seq:
                f()
    for _ in range(3):
        g.launch()
    torch.cuda.synchronize()
    with bench.event_timer(lib)() as tm:
        for _ in range(10):
            g.launch()
    us = tm.ms * 1e3 / (inner * 10)
    print('%-30s %.2f us per launch  %.2f TB/s' % (name, us, nbytes / us / 1e6))
