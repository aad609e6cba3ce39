"""CPU baseline of the decode path assembled from stock torch.nn modules.

TEST INFRASTRUCTURE ONLY (see oracle/tts_oracle.py header): imported by tests/ and by the
`cpu_baseline` leg of bench.py, never by the product path.

Why it exists next to tts_oracle.py: the oracle is a *functional* restatement (matmul + pointwise,
written for readability and for autograd in float64); it does not hit the fused ATen kernels the
reference hits (`nn.LSTMCell` -> `_thnn_fused_lstm_cell`/addmm, `nn.Conv1d`, `F.dropout`).  SURVEY.md
8d asks for the CPU reference on the GPU box to be "plain PyTorch-CPU nn modules assembled by the
build (same ATen kernels the reference would hit)".  This file is that assembly for
`Decoder.forward` in free-running inference (the headline region): ref src/module.py:140-214 (loop),
:216-288 (step), :320-340 (Prenet), :371-407 (Attention).  tests/test_oracle_golden.py checks it
against the outputs recorded from the real reference (tests/golden/tts_tiny_infer*.npz).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

Tensor = torch.Tensor


def _linear(w: Tensor, b: Tensor = None) -> nn.Linear:
    m = nn.Linear(w.shape[1], w.shape[0], bias=b is not None)
    m.weight.data.copy_(w)
    if b is not None:
        m.bias.data.copy_(b)
    return m


def _cell(W: Dict[str, Tensor], p: str) -> nn.LSTMCell:
    w_ih, w_hh = W[p + '.weight_ih'], W[p + '.weight_hh']
    c = nn.LSTMCell(w_ih.shape[1], w_hh.shape[1])
    for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
        getattr(c, n).data.copy_(W[p + '.' + n])
    return c


class NNDecoder(nn.Module):
    """Decoder.forward, tf_rate = 0 (free running), spkr_embed_mode 'adaIN', eval mode (LSTM dropouts off, the
    prenet dropout on: src/module.py:339)."""

    def __init__(self, W: Dict[str, Tensor], hp: dict, prefix: str = 'decoder.'):
        super().__init__()
        g = lambda k: W[prefix + k]
        self.r, self.n_mels, self.p_pre = hp['n_frames_per_step'], hp['n_mels'], hp['prenet_dropout']
        self.pre = nn.ModuleList()
        i = 0
        while (prefix + 'prenet.layers.%d.linear.weight' % i) in W:
            self.pre.append(_linear(g('prenet.layers.%d.linear.weight' % i)))
            i += 1
        self.query_rnn, self.dec_rnn = _cell(W, prefix + 'query_rnn'), _cell(W, prefix + 'dec_rnn')
        self.query_layer = _linear(g('attn.query_layer.linear.weight'))
        self.memory_layer = _linear(g('attn.memory_layer.linear.weight'))
        self.v = _linear(g('attn.v.linear.weight'))
        wc = g('attn.loc_conv.conv.weight')
        self.loc_conv = nn.Conv1d(wc.shape[1], wc.shape[0], wc.shape[2], padding=(wc.shape[2] - 1) // 2, bias=False)
        self.loc_conv.weight.data.copy_(wc)
        self.loc_linear = _linear(g('attn.loc_linear.linear.weight'))
        self.ada_mean = _linear(g('pseudo_latent_mean.weight'), g('pseudo_latent_mean.bias'))
        self.ada_std = _linear(g('pseudo_latent_std.0.weight'), g('pseudo_latent_std.0.bias'))
        self.proj = _linear(g('proj.linear.weight'), g('proj.linear.bias'))
        self.gate = _linear(g('gate_layer.linear.weight'), g('gate_layer.linear.bias'))
        self.eval()

    def prenet(self, x: Tensor) -> Tensor:
        for lin in self.pre:
            x = F.dropout(F.relu(lin(x)), p=self.p_pre, training=True)          # :337-339, never turns off
        return x

    def forward(self, memory: Tensor, max_frames: int, spkr_embed: Tensor, seed: int = 0):
        torch.manual_seed(seed)
        B, L, E = memory.shape
        Q, D = self.query_rnn.hidden_size, self.dec_rnn.hidden_size
        z = lambda *s: memory.new_zeros(*s)
        h_q, c_q, h_d, c_d = z(B, Q), z(B, Q), z(B, D), z(B, D)                   # :290-303
        w, w_cum, ctx = z(B, L), z(B, L), z(B, E)
        pm = self.memory_layer(memory)                                          # :306
        dec_in = self.prenet(z(B, self.r * self.n_mels))                        # :161,:183
        mels, aligns, stops = [], [], []
        for _ in range(max_frames // self.r):                                   # :168
            h_q, c_q = self.query_rnn(torch.cat([dec_in, ctx], dim=-1), (h_q, c_q))           # :227-228
            pq = self.query_layer(h_q).unsqueeze(1)                                           # :380
            loc = self.loc_linear(self.loc_conv(torch.stack([w, w_cum], dim=1)).transpose(1, 2))   # :384-385
            e = self.v(torch.tanh(pq + loc + pm)).squeeze(-1)                                 # :389-391
            w = F.softmax(e, dim=1)                                                           # :403
            ctx = torch.bmm(w.unsqueeze(1), memory).squeeze(1)                                # :405-406
            w_cum = w_cum + w                                                                 # :264
            adapted = F.relu(self.ada_std(spkr_embed)) * (h_q - self.ada_mean(spkr_embed))    # :267-269
            h_d, c_d = self.dec_rnn(torch.cat([ctx, adapted], dim=-1), (h_d, c_d))            # :275-277
            y = torch.cat([h_d, ctx], dim=-1)                                                 # :282-284
            mel = self.proj(y)                                                                # :285
            mels.append(mel.view(B, self.r, self.n_mels))
            aligns.append(w)
            stops.append(self.gate(y).repeat(1, self.r))                                      # :287
            dec_in = self.prenet(mel)                                                         # :192
        return torch.cat(mels, dim=1), torch.stack(aligns, dim=1), torch.cat(stops, dim=1)
