"""CPU oracle for the Tacotron-style TTS hot path of ttaoREtw/semi-tts.

TEST INFRASTRUCTURE ONLY.  This file is the parity yard-stick: a functional
restatement (fp32, torch CPU tensor arithmetic) of what the reference computes in
`src/module.py:53-622`, `src/tts.py:9-51`.  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it.  The product path
(`semi_tts_amd/`) never routes through this file and raises when the HIP library is
missing.

Parity pinning: the reference ships no tests/golden vectors, so
this oracle is pinned against outputs of the *reference itself* imported in the build
container (`tools/gen_golden.py` -> `tests/golden/*.npz`, checked by
`tests/test_oracle_golden.py`).

All weights are passed as a flat ``dict`` keyed exactly like the reference's
``state_dict`` (e.g. ``decoder.query_rnn.weight_ih``), so golden fixtures generated
from the reference load without renaming.

Every function cites the reference lines it follows as ``ref: file:line``.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Weights = Dict[str, Tensor]


# --------------------------------------------------------------------------- dropout
class DropoutSource:
    """Supplies dropout masks in the order the reference would draw them.

    mode 'off'   : every dropout is the identity (parity runs with p forced to 0)
    mode 'list'  : pops pre-recorded *scaled* masks (value 0 or 1/(1-p)), recorded by
                   wrapping ``F.dropout`` around the real reference (tools/gen_golden.py)
    mode 'rng'   : draws bernoulli masks from a torch generator
    Every mask handed out is appended to ``self.used`` so that a test can replay the
    exact same masks through the HIP path.
    """

    def __init__(self, mode: str = 'off', masks: Optional[Sequence[Tensor]] = None,
                 generator: Optional[torch.Generator] = None):
        assert mode in ('off', 'list', 'rng')
        self.mode = mode
        self.masks = list(masks) if masks is not None else []
        self.pos = 0
        self.gen = generator
        self.used: List[Tensor] = []

    def __call__(self, x: Tensor, p: float, training: bool) -> Tensor:
        # ref: F.dropout semantics -- identity when not training or p == 0
        if (not training) or p == 0.0 or self.mode == 'off':
            return x
        if self.mode == 'list':
            m = self.masks[self.pos]
            self.pos += 1
            assert m.shape == x.shape, (m.shape, x.shape)
        else:
            keep = torch.full_like(x, 1.0 - p)
            m = torch.bernoulli(keep, generator=self.gen) / (1.0 - p)
        self.used.append(m)
        return x * m


# --------------------------------------------------------------------------- primitives
def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """y = x W^T + b   (torch.nn.Linear; ref: src/module.py:500-522 with norm_type=None)"""
    y = x.matmul(w.t())
    if b is not None:
        y = y + b
    return y


def conv1d_cl(x: Tensor, w: Tensor, b: Optional[Tensor], pad: int) -> Tensor:
    """Conv1d on a channels-last activation.

    x (B,T,Cin), w (Cout,Cin,k) torch layout, stride 1, zero padding `pad` both sides.
    Returns (B, T + 2*pad - k + 1, Cout).   ref: torch.nn.Conv1d as used in
    src/module.py:480-497 (Conv1d wrapper) and :530 (BatchNormConv1d).
    Written as a sum over taps of shifted matmuls (the same decomposition the HIP
    implicit-GEMM uses) rather than calling F.conv1d, so the oracle is an independent
    statement of the arithmetic.
    """
    B, T, Cin = x.shape
    Cout, Cin2, k = w.shape
    assert Cin == Cin2
    xp = F.pad(x, (0, 0, pad, pad))
    Tout = T + 2 * pad - k + 1
    y = torch.zeros(B, Tout, Cout, dtype=x.dtype)
    for tap in range(k):
        y = y + xp[:, tap:tap + Tout, :].matmul(w[:, :, tap].t())
    if b is not None:
        y = y + b
    return y


def batchnorm_cl(x: Tensor, W: Weights, prefix: str, eps: float, momentum: float,
                 training: bool, stats_out: Optional[dict] = None) -> Tensor:
    """BatchNorm1d over channels-last x (B,T,C): statistics over (B,T).

    eval: (x - running_mean) / sqrt(running_var + eps) * gamma + beta
    train: batch mean / biased batch var for the output; running stats updated with the
    unbiased var (torch semantics).  ref: nn.BatchNorm1d at src/module.py:434 (encoder,
    default eps 1e-5, momentum 0.1) and :531 (CBHG, eps 1e-3, momentum 0.99).
    """
    g, beta = W[prefix + '.weight'], W[prefix + '.bias']
    if training:
        n = x.shape[0] * x.shape[1]
        mean = x.mean(dim=(0, 1))
        var_b = x.var(dim=(0, 1), unbiased=False)
        if stats_out is not None:
            var_u = var_b * (n / max(n - 1, 1))
            stats_out[prefix + '.running_mean'] = (1 - momentum) * W[prefix + '.running_mean'] + momentum * mean
            stats_out[prefix + '.running_var'] = (1 - momentum) * W[prefix + '.running_var'] + momentum * var_u
            stats_out[prefix + '.batch_mean'] = mean
            stats_out[prefix + '.batch_var'] = var_b
        return (x - mean) / torch.sqrt(var_b + eps) * g + beta
    mean, var = W[prefix + '.running_mean'], W[prefix + '.running_var']
    return (x - mean) / torch.sqrt(var + eps) * g + beta


def lstm_cell(x: Tensor, h: Tensor, c: Tensor, w_ih: Tensor, w_hh: Tensor,
              b_ih: Tensor, b_hh: Tensor) -> Tuple[Tensor, Tensor]:
    """torch.nn.LSTMCell: gate order (i,f,g,o), both biases summed.
    ref: nn.LSTMCell at src/module.py:127-128,133-134, called :228,:277."""
    H = h.shape[1]
    gates = x.matmul(w_ih.t()) + b_ih + h.matmul(w_hh.t()) + b_hh
    i = torch.sigmoid(gates[:, 0 * H:1 * H])
    f = torch.sigmoid(gates[:, 1 * H:2 * H])
    g = torch.tanh(gates[:, 2 * H:3 * H])
    o = torch.sigmoid(gates[:, 3 * H:4 * H])
    c2 = f * c + i * g
    h2 = o * torch.tanh(c2)
    return h2, c2


def lstm_layer(x: Tensor, W: Weights, prefix: str, reverse: bool, layer: int = 0) -> Tensor:
    """One direction of one layer of nn.LSTM(batch_first) over a full-length (unpacked) sequence,
    zero initial state.  ref: src/module.py:432-438,458-460 (lengths are ignored)."""
    sfx = '_l%d%s' % (layer, '_reverse' if reverse else '')
    w_ih, w_hh = W[prefix + '.weight_ih' + sfx], W[prefix + '.weight_hh' + sfx]
    b_ih, b_hh = W[prefix + '.bias_ih' + sfx], W[prefix + '.bias_hh' + sfx]
    B, L, _ = x.shape
    H = w_hh.shape[1]
    h = torch.zeros(B, H)
    c = torch.zeros(B, H)
    out = torch.zeros(B, L, H)
    order = range(L - 1, -1, -1) if reverse else range(L)
    for t in order:
        h, c = lstm_cell(x[:, t], h, c, w_ih, w_hh, b_ih, b_hh)
        out[:, t] = h
    return out


def gru_layer(x: Tensor, W: Weights, prefix: str, reverse: bool) -> Tensor:
    """One direction of nn.GRU(batch_first), gate order (r,z,n):
    r = s(Wir x + bir + Whr h + bhr); z likewise; n = tanh(Win x + bin + r*(Whn h + bhn));
    h' = (1-z)*n + z*h.   ref: nn.GRU at src/module.py:585-586, called :617."""
    sfx = '_l0_reverse' if reverse else '_l0'
    w_ih, w_hh = W[prefix + '.weight_ih' + sfx], W[prefix + '.weight_hh' + sfx]
    b_ih, b_hh = W[prefix + '.bias_ih' + sfx], W[prefix + '.bias_hh' + sfx]
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = torch.zeros(B, H)
    out = torch.zeros(B, T, H)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        gi = x[:, t].matmul(w_ih.t()) + b_ih
        gh = h.matmul(w_hh.t()) + b_hh
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1.0 - z) * n + z * h
        out[:, t] = h
    return out


# --------------------------------------------------------------------------- encoder
def encoder_forward(W: Weights, txt_embed: Tensor, prefix: str = 'encoder.',
                    training: bool = False, enc_dropout: float = 0.0,
                    drop: Optional[DropoutSource] = None, stats_out: Optional[dict] = None) -> Tensor:
    """Text encoder: n x [Conv1d(k, pad (k-1)//2, bias) -> BN -> ReLU -> Dropout] then
    a 1-layer BiLSTM, lengths ignored.   ref: src/module.py:410-462."""
    drop = drop or DropoutSource('off')
    x = txt_embed
    i = 0
    while (prefix + 'convs.%d.0.conv.weight' % i) in W:
        w = W[prefix + 'convs.%d.0.conv.weight' % i]
        b = W[prefix + 'convs.%d.0.conv.bias' % i]
        k = w.shape[2]
        x = conv1d_cl(x, w, b, (k - 1) // 2)                                   # :421-428
        x = batchnorm_cl(x, W, prefix + 'convs.%d.1' % i, 1e-5, 0.1, training, stats_out)  # :429
        x = torch.relu(x)                                                      # :430
        # :431 -- the reference drops on its (B, C, L) layout: a recorded mask is replayed through the same layout
        x = drop(x.transpose(1, 2), enc_dropout, training).transpose(1, 2)
        i += 1
    layer = 0
    while (prefix + 'lstm.weight_ih_l%d' % layer) in W:                         # enc_rnn_layer stacked BiLSTM layers (1 in the configs)
        fw = lstm_layer(x, W, prefix + 'lstm', False, layer)                    # :458-460
        bw = lstm_layer(x, W, prefix + 'lstm', True, layer)
        x = torch.cat([fw, bw], dim=-1)
        layer += 1
    return x


# --------------------------------------------------------------------------- decoder
def prenet_forward(W: Weights, x: Tensor, p: float, drop: DropoutSource,
                   prefix: str = 'decoder.prenet.', norm_type: Optional[str] = None, training: bool = False,
                   stats_out: Optional[dict] = None) -> Tensor:
    """Prenet: per layer relu(Linear_nobias(x)) then dropout with training=True ALWAYS.
    ref: src/module.py:320-340 (:339 'The dropout does NOT turn off').
    norm_type 'LayerNorm' / 'BatchNorm1d': the Linear wrapper normalises its output before the ReLU (src/module.py:508-521;
    BatchNorm1d sees a (B, T, C) input transposed and a (B, C) input as it is: statistics over every leading row in both cases;
    in training mode the running statistics move on -- `stats_out`, fed back through W by the caller for per-step use)."""
    i = 0
    while (prefix + 'layers.%d.linear.weight' % i) in W:
        x = linear(x, W[prefix + 'layers.%d.linear.weight' % i])
        pn = prefix + 'layers.%d.norm' % i
        if norm_type == 'LayerNorm':
            x = F.layer_norm(x, (x.shape[-1],), W[pn + '.weight'], W[pn + '.bias'], 1e-5)
        elif norm_type == 'BatchNorm1d':
            lead = x.shape[:-1]
            so = {} if (training and stats_out is None) else stats_out
            x = batchnorm_cl(x.reshape(1, -1, x.shape[-1]), W, pn, 1e-5, 0.1, training, so).reshape(*lead, -1)
            if training and so is not None:         # the next call of this layer (the next decode step) sees the updated statistics
                W[pn + '.running_mean'], W[pn + '.running_var'] = so[pn + '.running_mean'], so[pn + '.running_var']
        x = torch.relu(x)
        x = drop(x, p, True)
        i += 1
    return x


def attention_step(W: Weights, query: Tensor, memory: Tensor, processed_memory: Tensor,
                   w_prev: Tensor, w_cum: Tensor, prefix: str = 'decoder.attn.') -> Tuple[Tensor, Tensor]:
    """Location-sensitive attention, one step, no mask (mask=None at src/module.py:163).
    ref: src/module.py:371-407.
      pq   = W_q query                                        :380
      loc  = W_l conv1d(stack[w_prev, w_cum]; 2->F, k, pad (k-1)//2, no bias)^T   :384-385
      e    = v . tanh(pq + loc + pm)                          :389-391
      w    = softmax_L(e)                                     :403
      ctx  = w @ memory                                       :405-406
    """
    pq = linear(query, W[prefix + 'query_layer.linear.weight']).unsqueeze(1)        # (B,1,A)
    if (prefix + 'loc_conv.conv.weight') in W:                                       # loc_aware (:382-385)
        wc = W[prefix + 'loc_conv.conv.weight']                                      # (F,2,k), or (F,1,k) without use_summed_weights
        # attn_history: stack[w_prev, w_cum] or w_prev alone (:235-239)               (B,L,C) channels-last
        hist = torch.stack([w_prev, w_cum], dim=-1) if wc.shape[1] == 2 else w_prev.unsqueeze(-1)
        k = wc.shape[2]
        loc = conv1d_cl(hist, wc, None, (k - 1) // 2)                                # (B,L,F)
        loc = linear(loc, W[prefix + 'loc_linear.linear.weight'])                    # (B,L,A)
    else:
        loc = 0                                                                      # :386-387
    e = linear(torch.tanh(pq + loc + processed_memory), W[prefix + 'v.linear.weight']).squeeze(-1)
    w = torch.softmax(e, dim=1)
    ctx = torch.bmm(w.unsqueeze(1), memory).squeeze(1)
    return ctx, w


class DecoderState:
    """ref: Decoder.init_decoder_states, src/module.py:290-306"""

    def __init__(self, W: Weights, memory: Tensor, Q: int, D: int, prefix: str = 'decoder.'):
        B, L, E = memory.shape
        self.h_q = torch.zeros(B, Q)
        self.c_q = torch.zeros(B, Q)
        self.h_d = torch.zeros(B, D)
        self.c_d = torch.zeros(B, D)
        self.w = torch.zeros(B, L)
        self.w_cum = torch.zeros(B, L)
        self.ctx = torch.zeros(B, E)
        self.memory = memory
        self.pm = linear(memory, W[prefix + 'attn.memory_layer.linear.weight'])      # :306


def decode_one_step(W: Weights, st: DecoderState, dec_in: Tensor, spkr_embed: Tensor, r: int,
                    n_mels: int, q_drop: float, d_drop: float, training: bool,
                    drop: DropoutSource, prefix: str = 'decoder.', mode: str = 'adain',
                    pretrain: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
    """ref: Decoder.decode_one_step, src/module.py:216-288 (loc_aware=True, use_summed_weights=True; spkr_embed_mode
    'adaIN' is the shipped configuration, 'concat' / 'add' condition the memory the context is read from, :241-250;
    pretrain skips the attention, :238-240)."""
    B = dec_in.shape[0]
    xq = torch.cat([dec_in, st.ctx], dim=-1)                                          # :227
    h, c = lstm_cell(xq, st.h_q, st.c_q, W[prefix + 'query_rnn.weight_ih'], W[prefix + 'query_rnn.weight_hh'],
                     W[prefix + 'query_rnn.bias_ih'], W[prefix + 'query_rnn.bias_hh'])   # :228
    st.h_q = drop(h, q_drop, training)                                                # :230 (dropped h is the state)
    st.c_q = c                                                                        # :231
    if pretrain:                                                                      # :238-240
        ctx, w = torch.zeros_like(st.ctx), torch.zeros_like(st.w)
    else:
        mem = st.memory
        if mode == 'concat':                                                          # :243-245
            L = mem.shape[1]
            mem = linear(torch.cat([mem, spkr_embed.unsqueeze(1).repeat(1, L, 1)], dim=-1),
                         W[prefix + 'spkr_mem_proj.weight'], W[prefix + 'spkr_mem_proj.bias'])
        elif mode == 'add':                                                           # :246-247
            sp = linear(spkr_embed.unsqueeze(1), W[prefix + 'spkr_proj.weight'], W[prefix + 'spkr_proj.bias'])
            mem = linear(mem + sp, W[prefix + 'spkr_mem_proj.weight'], W[prefix + 'spkr_mem_proj.bias'])
        ctx, w = attention_step(W, st.h_q, mem, st.pm, st.w, st.w_cum, prefix + 'attn.')   # :256-261
    st.ctx = ctx                                                                      # :262
    st.w = w                                                                          # :263
    st.w_cum = w + st.w_cum                                                           # :264
    if mode == 'adain':
        # AdaIN speaker adaptation                                                     :267-269
        std = torch.relu(linear(spkr_embed, W[prefix + 'pseudo_latent_std.0.weight'], W[prefix + 'pseudo_latent_std.0.bias']))
        mean = linear(spkr_embed, W[prefix + 'pseudo_latent_mean.weight'], W[prefix + 'pseudo_latent_mean.bias'])
        adapted = std * (st.h_q - mean)
    else:
        adapted = st.h_q                                                              # :271-272
    xd = torch.cat([st.ctx, adapted], dim=-1)                                         # :275-276
    h, c = lstm_cell(xd, st.h_d, st.c_d, W[prefix + 'dec_rnn.weight_ih'], W[prefix + 'dec_rnn.weight_hh'],
                     W[prefix + 'dec_rnn.bias_ih'], W[prefix + 'dec_rnn.bias_hh'])       # :277
    st.h_d = drop(h, d_drop, training)                                                # :279
    st.c_d = c
    y = torch.cat([st.h_d, st.ctx], dim=-1)                                           # :282-284
    mel = linear(y, W[prefix + 'proj.linear.weight'], W[prefix + 'proj.linear.bias']).view(B, r, n_mels)  # :285-286
    stop = linear(y, W[prefix + 'gate_layer.linear.weight'], W[prefix + 'gate_layer.linear.bias']).repeat(1, r)  # :287
    return mel, st.w, stop


def decoder_forward(W: Weights, memory: Tensor, teacher: Union[int, Tensor], spkr_embed: Tensor,
                    hp: dict, tf_rate: float = 0.0, unpair_max_frame: Optional[int] = None,
                    training: bool = False, drop: Optional[DropoutSource] = None,
                    coin: Callable[[], float] = np.random.rand, prefix: str = 'decoder.', stats_out: Optional[dict] = None):
    """ref: Decoder.forward, src/module.py:140-214.

    hp: the `model.decoder.decoder` section of the YAML (n_frames_per_step, prenet_dropout,
    query_dropout, dec_dropout, drop_dec_in ...) plus 'n_mels'.
    `coin` replaces np.random.rand (two draws per step as in :190,:193).
    Returns (mel (B,T,n_mels), alignment (B,steps,L), stop (B,T)).
    """
    drop = drop or DropoutSource('off')
    r, n_mels = hp['n_frames_per_step'], hp['n_mels']
    p_pre, p_q, p_d = hp['prenet_dropout'], hp['query_dropout'], hp['dec_dropout']
    pn_type = hp.get('prenet_norm_type')
    if pn_type == 'BatchNorm1d' and training:
        W = dict(W)                # the running statistics of the prenet's BatchNorm move with every call
    B = memory.shape[0]
    Q = W[prefix + 'query_rnn.weight_hh'].shape[1]
    D = W[prefix + 'dec_rnn.weight_hh'].shape[1]
    partial_no_teacher = False
    teacher_bs = B
    if not isinstance(teacher, int):                                                  # :156-159
        teacher_bs = teacher.shape[0]
        partial_no_teacher = B != teacher_bs
    st = DecoderState(W, memory, Q, D, prefix)                                        # :162
    inference = tf_rate == 0.0                                                        # :166
    if inference:
        steps = teacher // r if isinstance(teacher, int) else teacher.shape[1]        # :168 (un-divided quirk)
    else:
        if partial_no_teacher:
            assert unpair_max_frame is not None
            steps = max(teacher.shape[1] // r, unpair_max_frame // r)                 # :172-173
        else:
            steps = teacher.shape[1] // r                                             # :177
        teacher = teacher.reshape(teacher.shape[0], -1, n_mels * r)                   # :178
        teacher = prenet_forward(W, teacher, p_pre, drop, prefix + 'prenet.', pn_type, training, stats_out)         # :179
    mels, aligns, stops = [], [], []
    dec_in = prenet_forward(W, torch.zeros(B, r * n_mels), p_pre, drop, prefix + 'prenet.', pn_type, training, stats_out)   # :161,:183
    for t in range(steps):
        mel, al, stop = decode_one_step(W, st, dec_in, spkr_embed, r, n_mels, p_q, p_d, training, drop, prefix,
                                        hp.get('spkr_embed_mode', 'adaIN').lower(), bool(hp.get('pretrain', False)))
        mels.append(mel)
        aligns.append(al)
        stops.append(stop)
        if inference or (coin() > tf_rate):                                           # :190
            dec_in = prenet_forward(W, mel.reshape(B, r * n_mels), p_pre, drop, prefix + 'prenet.', pn_type, training, stats_out)   # :192
        elif coin() < hp.get('drop_dec_in', 0.0):                                     # :193
            dec_in = teacher.mean(dim=1)                                              # :194
            if partial_no_teacher:
                own = mel[teacher_bs:].reshape(-1, r * n_mels)
                dec_in = torch.cat([dec_in, prenet_forward(W, own, p_pre, drop, prefix + 'prenet.', pn_type, training, stats_out)], dim=0)
        else:
            take = min(t, teacher.shape[1] - 1)                                       # :201
            dec_in = teacher[:, take, :]                                              # :202
            if partial_no_teacher:                                                    # :204-206
                own = mel[teacher_bs:].reshape(-1, r * n_mels)
                dec_in = torch.cat([dec_in, prenet_forward(W, own, p_pre, drop, prefix + 'prenet.', pn_type, training, stats_out)], dim=0)
    mel_out = torch.cat(mels, dim=1)                                                  # :209
    align = torch.stack(aligns).transpose(0, 1)                                       # :211
    stop_out = torch.cat(stops, dim=1)                                                # :213
    return mel_out, align, stop_out


# --------------------------------------------------------------------------- CBHG postnet
def cbhg_forward(W: Weights, inputs: Tensor, prefix: str = 'postnet.0.', training: bool = False,
                 stats_out: Optional[dict] = None) -> Tensor:
    """CBHG(n_mels, K): conv bank (k=1..K, pad k//2, trimmed to T, conv->ReLU->BN) ->
    MaxPool1d(2,1,pad 1)[:T] -> conv proj (k3; ReLU+BN, then BN only) -> Linear(no bias)
    -> + inputs -> 4 x Highway -> BiGRU.      ref: src/module.py:558-622."""
    B, T, C = inputs.shape
    outs = []
    k = 1
    while (prefix + 'conv1d_banks.%d.conv1d.weight' % (k - 1)) in W:
        p = prefix + 'conv1d_banks.%d' % (k - 1)
        # even k yields T+1 positions; ReLU and BN (hence training-mode batch statistics)
        # see all T+1 of them, the trim to T happens afterwards in CBHG.forward   :597-598
        y = conv1d_cl(inputs, W[p + '.conv1d.weight'], None, k // 2)
        y = torch.relu(y)                                                             # :535-536 (activation first)
        y = batchnorm_cl(y, W, p + '.bn', 1e-3, 0.99, training, stats_out)            # :537, eps/momentum :531
        outs.append(y[:, :T])
        k += 1
    x = torch.cat(outs, dim=-1)                                                       # :596-599
    # MaxPool1d(kernel 2, stride 1, padding 1)[:T]  ==  max(x[t-1], x[t]) with x[-1] = -inf   :600
    prev = torch.cat([torch.full((B, 1, x.shape[2]), -float('inf')), x[:, :-1]], dim=1)
    x = torch.maximum(prev, x)
    i = 0
    n_proj = 0
    while (prefix + 'conv1d_projs.%d.conv1d.weight' % n_proj) in W:
        n_proj += 1
    for i in range(n_proj):                                                           # :602-603
        p = prefix + 'conv1d_projs.%d' % i
        x = conv1d_cl(x, W[p + '.conv1d.weight'], None, 1)
        if i < n_proj - 1:                                                            # :576 activations = [relu]*(n-1)+[None]
            x = torch.relu(x)
        x = batchnorm_cl(x, W, p + '.bn', 1e-3, 0.99, training, stats_out)
    x = linear(x, W[prefix + 'pre_highway_proj.weight'])                              # :607
    x = x + inputs                                                                    # :609
    i = 0
    while (prefix + 'highways.%d.H.weight' % i) in W:                                 # :610-611, :541-555
        p = prefix + 'highways.%d' % i
        Hh = torch.relu(linear(x, W[p + '.H.weight'], W[p + '.H.bias']))
        Tt = torch.sigmoid(linear(x, W[p + '.T.weight'], W[p + '.T.bias']))
        x = Hh * Tt + x * (1.0 - Tt)
        i += 1
    fw = gru_layer(x, W, prefix + 'gru', reverse=False)                               # :617
    bw = gru_layer(x, W, prefix + 'gru', reverse=True)
    return torch.cat([fw, bw], dim=-1)


def postnet_forward(W: Weights, mel: Tensor, training: bool = False, stats_out: Optional[dict] = None) -> Tensor:
    """Sequential(CBHG(n_mels, K=8), Linear(2*n_mels, linear_dim)).   ref: src/tts.py:31-34"""
    y = cbhg_forward(W, mel, 'postnet.0.', training, stats_out)
    return linear(y, W['postnet.1.weight'], W['postnet.1.bias'])


def conv_postnet_forward(W: Weights, x: Tensor, prefix: str = '', training: bool = False,
                         p_drop: float = 0.0, drop: Optional[DropoutSource] = None) -> Tensor:
    """The Tacotron-2 5-conv `Postnet` class (defined, imported, never constructed by any
    config).  n x [Conv1d(k,pad (k-1)//2) -> BN -> tanh (identity on the last) -> Dropout].
    ref: src/module.py:53-82."""
    drop = drop or DropoutSource('off')
    n = 0
    while (prefix + 'convs.%d.0.conv.weight' % n) in W:
        n += 1
    for i in range(n):
        w, b = W[prefix + 'convs.%d.0.conv.weight' % i], W[prefix + 'convs.%d.0.conv.bias' % i]
        x = conv1d_cl(x, w, b, (w.shape[2] - 1) // 2)
        x = batchnorm_cl(x, W, prefix + 'convs.%d.1' % i, 1e-5, 0.1, training)
        if i < n - 1:
            x = torch.tanh(x)
        x = drop(x, p_drop, training)
    return x


# --------------------------------------------------------------------------- Tacotron2
def tacotron2_forward(W: Weights, txt_embed: Tensor, teacher: Union[int, Tensor], spkr_embed: Tensor,
                      hp: dict, tf_rate: float = 0.0, unpair_max_frame: Optional[int] = None,
                      training: bool = False, drop: Optional[DropoutSource] = None,
                      coin: Callable[[], float] = np.random.rand, stats_out: Optional[dict] = None):
    """ref: Tacotron2.forward, src/tts.py:36-51.  `W` holds the `tts.`-relative keys
    (encoder.*, decoder.*, postnet.*).  hp = decoder hyper-parameters + 'n_mels' +
    'enc_dropout'.  separate_postnet only changes gradients (detach), not values.
    Returns (mel_pred, linear_pred, alignment, stop)."""
    drop = drop or DropoutSource('off')
    enc = encoder_forward(W, txt_embed, 'encoder.', training, hp.get('enc_dropout', 0.0), drop, stats_out)   # :44
    mel, align, stop = decoder_forward(W, enc, teacher, spkr_embed, hp, tf_rate, unpair_max_frame,
                                       training, drop, coin, 'decoder.', stats_out)                        # :45-46
    lin = postnet_forward(W, mel, training, stats_out) if 'postnet.1.weight' in W else None                 # :47-50
    return mel, lin, align, stop


# --------------------------------------------------------------------------- harness pieces (H1)
def freq_loss(pred: Tensor, label: Tensor, sample_rate: int, n_mels: int, loss: str = 'mse',
              differential_loss: bool = True, emphasize_linear_low: bool = True) -> Tensor:
    """ref: src/util.py:80-126 (p = 1)."""
    crit = F.l1_loss if loss == 'l1' else F.mse_loss
    dim = pred.shape[-1]
    out = crit(pred, label)
    if dim != n_mels and emphasize_linear_low:
        n_low = int(dim * (3000 / (sample_rate / 2)))
        out = 0.5 * out + 0.5 * crit(pred[:, :, :n_low], label[:, :, :n_low])
    if dim == n_mels and differential_loss:
        out = out + 0.5 * crit(pred[:, 1:] - pred[:, :-1], label[:, 1:] - label[:, :-1])
    return out


def lr_schedule(step: int, lr: float, kind: str) -> float:
    """ref: src/optim.py:21-31 ('warmup' 4000 / 'decay' 1000; 'fixed' otherwise)."""
    if kind not in ('warmup', 'decay'):
        return lr
    ws = 4000.0 if kind == 'warmup' else 1000.0
    return float(lr * ws ** 0.5 * min((step + 1) * ws ** -1.5, (step + 1) ** -0.5))


def padded_frames(T: int, r: int) -> int:
    """ref: bin/train_vqvae.py:43-46 -- pad r - T % r frames (at least one)."""
    return T + (r - T % r)
