"""CPU oracle for the CTC speech encoder, the producer of the VQ input.

TEST INFRASTRUCTURE ONLY (same rules as tts_oracle.py: only tests/, smoke() and bench's cpu_baseline may
import it).  Functional fp32 restatement of `CTC.forward` (src/asr.py:5-64) and `ConvLayer.forward`
(src/module.py:627-648) on weights keyed like the reference's state_dict (`layer0.conv.weight`,
`layer0.bn.running_mean`, `rnn.weight_ih_l1_reverse`, `postnet.weight`, ...).
Pinned by tests/golden/asr_tiny_{eval,train}.npz, recorded from the real reference (tools/gen_golden.py).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .tts_oracle import DropoutSource, lstm_cell

Tensor = torch.Tensor


def conv_layer(W: Dict[str, Tensor], x: Tensor, prefix: str, stride: int, residual: bool, batch_norm: bool,
               activation: str, training: bool, p_drop: float, drop: DropoutSource, stats_out: Optional[dict]) -> Tensor:
    """x (B,T,C) channels-last.  Conv1d(k, stride, padding 1 (0 when k == 1)) -> BN -> act -> (+x) -> dropout.
    ref: src/module.py:627-648"""
    w, b = W[prefix + '.conv.weight'], W[prefix + '.conv.bias']
    k = w.shape[2]
    feat = F.conv1d(x.transpose(1, 2), w, b, stride=stride, padding=1 if k != 1 else 0).transpose(1, 2)   # :633-634,:640
    if batch_norm:                                                                                       # :641-642
        g, beta = W[prefix + '.bn.weight'], W[prefix + '.bn.bias']
        if training:
            n = feat.shape[0] * feat.shape[1]
            mean = feat.mean(dim=(0, 1))
            var_b = feat.var(dim=(0, 1), unbiased=False)
            if stats_out is not None:
                stats_out[prefix + '.bn.running_mean'] = 0.9 * W[prefix + '.bn.running_mean'] + 0.1 * mean
                stats_out[prefix + '.bn.running_var'] = 0.9 * W[prefix + '.bn.running_var'] + 0.1 * var_b * (n / max(n - 1, 1))
        else:
            mean, var_b = W[prefix + '.bn.running_mean'], W[prefix + '.bn.running_var']
        feat = (feat - mean) / torch.sqrt(var_b + 1e-5) * g + beta
    feat = getattr(torch, activation.lower())(feat)                                                       # :643
    if residual:                                                                                          # :644-645
        feat = feat + x
    return drop(feat, p_drop, training)                                                                   # :646


def lstm_layer_n(x: Tensor, W: Dict[str, Tensor], prefix: str, layer: int, reverse: bool) -> Tensor:
    sfx = '_l%d%s' % (layer, '_reverse' if reverse else '')
    w_ih, w_hh = W[prefix + '.weight_ih' + sfx], W[prefix + '.weight_hh' + sfx]
    b_ih, b_hh = W[prefix + '.bias_ih' + sfx], W[prefix + '.bias_hh' + sfx]
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h, c = torch.zeros(B, H), torch.zeros(B, H)
    out = torch.zeros(B, T, H)
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        h, c = lstm_cell(x[:, t], h, c, w_ih, w_hh, b_ih, b_hh)
        out[:, t] = h
    return out


def ctc_forward(W: Dict[str, Tensor], x: Tensor, cfg: dict, training: bool = False,
                drop: Optional[DropoutSource] = None, stats_out: Optional[dict] = None, prefix: str = '') -> Tensor:
    """x (B,T,n_mels) -> (B, T / time_reduce_factor, out_dim).  cfg = the `model.encoder` YAML section.
    ref: CTC.forward src/asr.py:46-64.  Dropout between the LSTM layers (nn.LSTM(dropout=...)) is drawn inside
    torch and cannot be replayed: parity cases use dropout 0 in training mode."""
    drop = drop or DropoutSource('off')
    for l, (s, r) in enumerate(zip(cfg['stride'], cfg['residual'])):                                       # :51-52
        x = conv_layer(W, x, prefix + 'layer%d' % l, s, bool(r), cfg['batch_norm'], cfg['activation'], training,
                       cfg['dropout'], drop, stats_out)
    for layer in range(cfg['rnn_layers']):                                                                 # :56
        if cfg['rnn_bid']:                                                                                 # :35-37
            x = torch.cat([lstm_layer_n(x, W, prefix + 'rnn', layer, False), lstm_layer_n(x, W, prefix + 'rnn', layer, True)], -1)
        else:
            x = lstm_layer_n(x, W, prefix + 'rnn', layer, False)
        if layer + 1 < cfg['rnn_layers']:
            x = drop(x, cfg['dropout'], training)       # nn.LSTM inter-layer dropout (identity in the parity cases)
    if cfg['layer_norm']:                                                                                  # :57-58
        x = torch.nn.functional.layer_norm(x, (x.shape[-1],), W[prefix + 'norm_layer.weight'], W[prefix + 'norm_layer.bias'], 1e-5)
    x = drop(x, cfg['dropout'], training)                                                                  # :62
    return x.matmul(W[prefix + 'postnet.weight'].t()) + W[prefix + 'postnet.bias']


def asr_postnet_forward(W: Dict[str, Tensor], x: Tensor, training: bool = False, drop: Optional[DropoutSource] = None,
                        prefix: str = '') -> Tensor:
    """ASRPostnet: 2-layer BiLSTM (inter-layer dropout 0.5) -> dropout 0.5 -> Linear -> log_softmax.   ref: src/asr.py:67-80"""
    drop = drop or DropoutSource('off')
    for layer in range(2):
        x = torch.cat([lstm_layer_n(x, W, prefix + 'rnn', layer, False), lstm_layer_n(x, W, prefix + 'rnn', layer, True)], -1)
        x = drop(x, 0.5, training)          # after layer 0: nn.LSTM(dropout=0.5); after layer 1: self.dropout         :78-79
    x = x.matmul(W[prefix + 'linear.weight'].t()) + W[prefix + 'linear.bias']
    return torch.log_softmax(x, dim=-1)                                                                    # :80
