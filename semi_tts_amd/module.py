"""Host-side mirror of the reference's TTS building blocks (src/module.py:53-622) on the
MI355X HIP path.

Same class names, constructor arguments, forward signatures and state_dict keys as the
reference, so `load_state_dict` of a reference checkpoint works and callers
(`Tacotron2`, `VQVAE.text_to_speech`) do not change.  torch.nn modules appear here only as
parameter containers (they give the reference's parameter names and initialisation); none
of their `forward`s is ever called -- every forward below goes through libsemitts_hip.so
via `ops` and raises on CPU tensors.

Activations are kept channels-last (B, T, C) end to end (the reference transposes to
(B, C, T) around every Conv1d; the HIP conv is an implicit GEMM over channels-last rows).
"""
import ctypes as C

import numpy as np
import os
import torch
import torch.nn as nn

from . import ops
from . import _lib
from . import autograd as AG
from ._lib import StDecoderWeights, StDecoderDims, StDecoderIO, check


_ONES = {}


def _scaled_mask(shape, p, device):
    """bernoulli(1-p)/(1-p) dropout mask (device RNG; the values the reference's F.dropout multiplies by): F.dropout of a cached tensor of
    ones -- ONE launch (bernoulli_ + div_ were two)"""
    n = int(torch.Size(shape).numel())
    key = (str(device), n)
    ones = _ONES.get(key)
    if ones is None:
        if len(_ONES) > 64:
            _ONES.clear()
        ones = _ONES[key] = torch.ones(n, device=device, dtype=torch.float32)
    return torch.nn.functional.dropout(ones, p, training=True).view(shape)


def _scaled_masks(shapes, p, device):
    """several dropout masks of one rate from ONE draw (two launches for all of them instead of two each)"""
    sizes = [int(torch.Size(s).numel()) for s in shapes]
    flat = _scaled_mask((sum((n + 3) // 4 * 4 for n in sizes),), p, device)         # (16-byte aligned pieces)
    out, off = [], 0
    for shp, n in zip(shapes, sizes):
        out.append(flat[off:off + n].view(*shp))
        off += (n + 3) // 4 * 4
    return out


# ----------------------------------------------------------------------------- thin parameter wrappers
class Conv1d(nn.Module):
    """ref: src/module.py:480-497 (xavier-uniform init with `w_init_gain`)"""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, dilation=1,
                 bias=True, w_init_gain='linear'):
        super().__init__()
        if padding is None:
            assert kernel_size % 2 == 1, 'kernel_size should be odd if no given padding.'
            padding = (dilation * (kernel_size - 1)) // 2
        assert stride == 1 and dilation == 1, 'the HIP conv1d supports stride 1, dilation 1 (all the path uses)'
        self.padding = padding
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                              dilation=dilation, bias=bias)
        nn.init.xavier_uniform_(self.conv.weight, gain=nn.init.calculate_gain(w_init_gain))

    def forward(self, x_cl, **epilogue):
        """x_cl (B,T,Cin) channels-last -> (B,T',Cout)"""
        return ops.gemm(x_cl, self.conv.weight, pad=self.padding, bias=self.conv.bias, **epilogue)


class Linear(nn.Module):
    """ref: src/module.py:500-522"""

    def __init__(self, in_dim, out_dim, bias=True, w_init_gain='linear', norm_type=None):
        super().__init__()
        self.linear = nn.Linear(in_dim, out_dim, bias=bias)
        nn.init.xavier_uniform_(self.linear.weight, gain=nn.init.calculate_gain(w_init_gain))
        self.apply_norm = norm_type is not None
        if self.apply_norm:                                                              # ref: src/module.py:508-513
            assert norm_type in ['LayerNorm', 'BatchNorm1d']
            self.norm_type = norm_type
            self.norm = getattr(nn, norm_type)(out_dim)

    def _forward_norm(self, x, act, mask):
        """Linear -> LayerNorm / BatchNorm1d -> act -> mask (src/module.py:515-521; the caller's ReLU + dropout, :337-339).
        BatchNorm1d sees the channels of a (B, T, C) input after a transpose and of a (B, C) input directly: in both cases the
        statistics run over every leading row, which is how the rows-flattened kernels take it."""
        lead = x.shape[:-1]
        grad = self.training and torch.is_grad_enabled()
        if grad:
            y = AG.linear(x, self.linear.weight, self.linear.bias)
        else:
            x2 = x.reshape(-1, x.shape[-1])
            y = (ops.linear_small(x2, self.linear.weight, self.linear.bias, None, None) if x2.shape[0] <= 64
                 else ops.gemm(x2, self.linear.weight, bias=self.linear.bias))
        y2 = y.reshape(-1, y.shape[-1])
        nm = self.norm
        if self.norm_type == 'LayerNorm':
            y2 = AG.layer_norm(y2, nm) if grad else ops.layer_norm(y2, nm.weight, nm.bias, nm.eps)
            if act == 'relu':
                y2 = torch.relu(y2)
            elif act is not None:
                raise NotImplementedError(act)
        elif grad:
            y2 = AG.batch_norm_train(y2, nm, act)
        else:
            N = y2.shape[1]
            if self.training:       # batch statistics (+ running-stat update) without autograd
                mean, var = ops.bn_stats(y2, 0, N, nm.running_mean, nm.running_var, nm.momentum, nm.num_batches_tracked)
            else:
                mean, var = nm.running_mean, nm.running_var
            y2 = y2.contiguous()
            ops.bn_apply(y2, 0, N, mean, var, nm.weight, nm.bias, nm.eps, act)
        if mask is not None:
            y2 = y2 * mask.reshape(-1, mask.shape[-1])
        return y2.view(*lead, -1)

    def forward(self, x, act=None, mask=None):
        if self.apply_norm:
            return self._forward_norm(x, act, mask)
        if self.training:            # differentiable path (same kernels + saved tensors)
            return AG.linear(x, self.linear.weight, self.linear.bias, act, mask)
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        m2 = mask.reshape(-1, mask.shape[-1]) if mask is not None else None
        if x2.shape[0] <= 64:
            y = ops.linear_small(x2, self.linear.weight, self.linear.bias, act, m2)
        else:
            y = ops.gemm(x2, self.linear.weight, bias=self.linear.bias, act_pre=act, mask=m2)
        return y.view(*lead, -1)


# ----------------------------------------------------------------------------- batch norm helper
def _conv_bn_act(x, weight, bias, bn, pad, eps, momentum, order, training, Tout=None, out=None, coff=0,
                 pool_prev=False, stats_Tout=None, collect=None):
    """conv1d followed by {BN, act} in the order the reference applies them.
    order 'bn_relu' (encoder, module.py:429-430), 'relu_bn' (BatchNormConv1d with activation,
    :535-537), 'bn' (no activation), 'bn_tanh' (Postnet class).  Eval mode folds everything into
    the GEMM epilogue; training mode needs the batch statistics of the pre-BN tensor first."""
    pre = 'relu' if order == 'relu_bn' else None
    post = {'bn_relu': 'relu', 'bn_tanh': 'tanh'}.get(order)
    if not training:
        return ops.gemm(x, weight, out, pad=pad, Tout=Tout, coff=coff, bias=bias, act_pre=pre,
                        bn=(bn.running_mean, bn.running_var, bn.weight, bn.bias), bn_eps=eps, act_post=post,
                        pool_prev=pool_prev, collect=collect)
    assert collect is None, 'batched launches: inference only'
    # training: differentiable conv -> BatchNorm with batch statistics.  For the even-k bank convs the
    # statistics run over stats_Tout = T+1 positions and the extra one is trimmed afterwards (module.py:597-598)
    assert out is None, 'the training path returns new tensors (autograd owns them)'
    To = stats_Tout if stats_Tout is not None else Tout
    y = AG.conv(x, weight, bias, pad=pad, Tout=To, act=pre, pool_prev=pool_prev)
    y = AG.batch_norm_train(y, bn, post)
    if Tout is not None and To != Tout:
        y = y[:, :Tout]
    return y


# ----------------------------------------------------------------------------- encoder
class Encoder(nn.Module):
    """Tacotron2 text encoder: n x [Conv1d k5 + BN + ReLU + Dropout] -> BiLSTM, lengths ignored.
    ref: src/module.py:410-462"""

    def __init__(self, in_dim, enc_embed_dim, enc_n_conv, enc_rnn_layer, enc_kernel_size, enc_dropout=0.5):
        super().__init__()
        in_size = [in_dim] + [enc_embed_dim] * (enc_n_conv - 1)
        self.enc_embed_dim = enc_embed_dim
        self.enc_dropout = enc_dropout
        self.convs = nn.ModuleList([
            nn.Sequential(Conv1d(din, enc_embed_dim, kernel_size=enc_kernel_size, stride=1,
                                 padding=(enc_kernel_size - 1) // 2, dilation=1, w_init_gain='relu'),
                          nn.BatchNorm1d(enc_embed_dim), nn.ReLU(), nn.Dropout(enc_dropout))
            for din in in_size])
        self.enc_rnn_layer = enc_rnn_layer
        self.lstm = nn.LSTM(input_size=enc_embed_dim, hidden_size=enc_embed_dim // 2, num_layers=enc_rnn_layer,
                            batch_first=True, bidirectional=True)

    def forward(self, txt_embed, input_lengths=None, _masks=None):
        """_masks: optional list of scaled dropout masks (B, L, C), one per conv block (tests replay the reference's draws)"""
        x = txt_embed.contiguous()
        for i, blk in enumerate(self.convs):
            conv, bn = blk[0], blk[1]
            x = _conv_bn_act(x, conv.conv.weight, conv.conv.bias, bn, conv.padding, bn.eps, bn.momentum,
                             'bn_relu', self.training)
            if self.training and self.enc_dropout > 0:                                   # nn.Dropout of the block, :431
                x = x * (_masks[i] if _masks is not None else _scaled_mask(x.shape, self.enc_dropout, x.device))
        B, L, _ = x.shape
        H = self.lstm.hidden_size
        for layer in range(self.enc_rnn_layer):            # (nn.LSTM stacks the layers; no inter-layer dropout: module.py:432-438)
            g = lambda n, rev=False: getattr(self.lstm, '%s_l%d%s' % (n, layer, '_reverse' if rev else ''))
            if self.training:
                xp_f, xp_b = AG.conv_group(x, [g('weight_ih'), g('weight_ih', True)], [0, 0], [None, None],
                                           biases=[g('bias_ih'), g('bias_ih', True)])       # both directions: one launch
                x = AG.bilstm(xp_f, xp_b, g('weight_hh'), g('bias_hh'), g('weight_hh', True), g('bias_hh', True))
                continue
            out = torch.empty(B, L, 2 * H, device=x.device, dtype=torch.float32)
            jobs = []
            xp_f = ops.gemm(x, g('weight_ih'), bias=g('bias_ih'), collect=jobs)          # (B, L, 4H), all time steps at once;
            xp_b = ops.gemm(x, g('weight_ih', True), bias=g('bias_ih', True), collect=jobs)      # both directions in one launch
            ops.gemm_flush(jobs)
            # the two directions advance together: the whole layer as ONE launch where the shape allows (H % 64 == 0: every shipped
            # config), else one launch per time step for both
            ops.lstm_seq2(xp_f, xp_b, g('weight_hh'), g('weight_hh', True), g('bias_hh'), g('bias_hh', True), out)
            x = out
        return x


# ----------------------------------------------------------------------------- decoder pieces
class Prenet(nn.Module):
    """2 x [Linear(no bias) -> ReLU -> dropout(always on)].  ref: src/module.py:320-340"""

    def __init__(self, in_dim, hidden_dim=(256, 256), apply_dropout=0.5, norm_type=None):
        super().__init__()
        hidden_dim = list(hidden_dim)
        dims = [in_dim] + hidden_dim[:-1]
        self.layers = nn.ModuleList([Linear(din, dout, bias=False, norm_type=norm_type)
                                     for din, dout in zip(dims, hidden_dim)])
        self.apply_dropout = apply_dropout

    def forward(self, x, masks=None):
        """masks: optional list of scaled masks, one per layer (drawn on device when None)"""
        if masks is None and self.apply_dropout > 0:       # (all layers' masks from one draw)
            masks = _scaled_masks([x.shape[:-1] + (layer.linear.out_features,) for layer in self.layers], self.apply_dropout, x.device)
        for i, layer in enumerate(self.layers):
            x = layer(x, act='relu', mask=masks[i] if masks is not None else None)
        return x


class Attention(nn.Module):
    """Location-sensitive attention parameters.  ref: src/module.py:343-407.
    The per-step computation is the fused kernel st_attn_step_fwd (see Decoder)."""

    def __init__(self, query_dim, memory_dim, hidden_dim, n_location_filters, location_kernel_size,
                 loc_aware, use_summed_weights):
        super().__init__()
        self.query_layer = Linear(query_dim, hidden_dim, bias=False, w_init_gain='tanh')
        self.memory_layer = Linear(memory_dim, hidden_dim, bias=False, w_init_gain='tanh')
        self.v = Linear(hidden_dim, 1, bias=False)
        self.loc_aware = loc_aware
        self.use_summed_weights = use_summed_weights
        self._dims = (hidden_dim, n_location_filters, location_kernel_size)
        if loc_aware:                                                                    # ref: src/module.py:358-366
            self.loc_conv = Conv1d(in_channels=2 if use_summed_weights else 1, out_channels=n_location_filters,
                                   kernel_size=location_kernel_size, bias=False, stride=1, dilation=1)
            self.loc_linear = Linear(n_location_filters, hidden_dim, bias=False, w_init_gain='tanh')

    def location_weights(self):
        """(W_c (F, 2, K), W_l (A, F)) as the kernels take them.  The kernels always convolve [w_prev ; w_cum]: without
        use_summed_weights the cumulative channel gets zero taps (the conv over w_prev alone, src/module.py:239), without
        loc_aware both are zero (processed_loc_feat = 0, :386-387).  Differentiable in the real parameters (torch.cat)."""
        A, F_, K = self._dims
        if not self.loc_aware:
            dev = self.v.linear.weight.device
            return torch.zeros(F_, 2, K, device=dev), torch.zeros(A, F_, device=dev)
        wc = self.loc_conv.conv.weight
        if not self.use_summed_weights:
            wc = torch.cat([wc, torch.zeros_like(wc)], dim=1)
        return wc, self.loc_linear.linear.weight

    def process_memory(self, memory):
        return ops.gemm(memory, self.memory_layer.linear.weight)

    def forward(self, query, memory, processed_memory, attn_history, mask=None):
        """Single step with the reference's signature: attn_history (B,2,L) = stack[prev, cum]."""
        assert mask is None, 'the reference never passes a mask (module.py:163)'
        B, L, E = memory.shape
        pq = ops.linear_small(query, self.query_layer.linear.weight)
        hist = attn_history.contiguous()
        w_prev = hist[:, 0].contiguous()
        w_cum = hist[:, 1].contiguous() if hist.shape[1] > 1 else torch.zeros_like(w_prev)      # (B,1,L) without use_summed_weights
        w = torch.empty(B, L, device=memory.device, dtype=torch.float32)
        w_cum_new = torch.empty_like(w)
        ctx = torch.empty(B, E, device=memory.device, dtype=torch.float32)
        wc, wl = self.location_weights()
        ops.attn_step(pq, processed_memory, memory, w_prev, w_cum, w, w_cum_new, wc.detach().contiguous(), wl.detach().contiguous(),
                      self.v.linear.weight, ctx)
        return ctx, w


def plan_decode(teacher_is_int, teacher_frames, teacher_bs, B, r, tf_rate, drop_dec_in, unpair_max_frame,
                coin=None):
    """Host-side step-count and next-input policy of Decoder.forward (src/module.py:156-206).

    Returns (steps, step_src) with step_src[t] = source of the decoder input of step t+1:
    -1 prenet(own output), -2 teacher mean, k >= 0 teacher frame k (rows >= teacher_bs always
    feed back their own output).  `coin` is drawn exactly as the reference draws np.random.rand
    (one or two draws per step, including after the last step)."""
    coin = coin or np.random.rand          # looked up at call time, like the reference does
    inference = tf_rate == 0.0
    partial = (not teacher_is_int) and B != teacher_bs
    if inference:
        steps = teacher_frames // r if teacher_is_int else teacher_frames       # :168 (un-divided quirk)
        t_groups = None
    else:
        assert not teacher_is_int, 'teacher forcing needs a teacher tensor'
        t_groups = teacher_frames // r
        if partial:
            assert unpair_max_frame is not None
            steps = max(t_groups, unpair_max_frame // r)                          # :172-173
        else:
            steps = t_groups                                                      # :177
    src = []
    for t in range(steps):
        if inference or (coin() > tf_rate):                                       # :190
            src.append(-1)
        elif coin() < drop_dec_in:                                                # :193
            src.append(-2)
        else:
            src.append(min(t, t_groups - 1))                                      # :201
    return steps, src


class Decoder(nn.Module):
    """Tacotron2 decoder: prenet, query LSTM, location-sensitive attention, AdaIN speaker
    adaptation, decoder LSTM, r-frame projection + stop gate.  ref: src/module.py:85-317"""

    def __init__(self, n_mels, n_frames_per_step, enc_embed_dim, spkr_embed_dim, prenet_dim, prenet_dropout,
                 query_rnn_dim, dec_rnn_dim, query_dropout, dec_dropout, attn_dim, n_location_filters,
                 location_kernel_size, loc_aware, use_summed_weights, drop_dec_in, prenet_norm_type=None,
                 pretrain=False, spkr_embed_mode='adaIN'):
        super().__init__()
        self.n_mels = n_mels
        self.n_frames_per_step = n_frames_per_step
        self.enc_embed_dim = enc_embed_dim
        self.spkr_embed_dim = spkr_embed_dim
        self.query_rnn_dim = query_rnn_dim
        self.dec_rnn_dim = dec_rnn_dim
        self.prenet_dropout = prenet_dropout
        self.prenet_dim = prenet_dim
        self.query_dropout = nn.Dropout(query_dropout)
        self.dec_dropout = nn.Dropout(dec_dropout)
        self.attn_dim = attn_dim
        self.n_location_filters = n_location_filters
        self.location_kernel_size = location_kernel_size
        self.pretrain = pretrain
        self.loc_aware = loc_aware
        self.use_summed_weights = use_summed_weights
        self.drop_dec_in = drop_dec_in
        self.prenet_norm_type = prenet_norm_type
        self.spkr_embed_mode = spkr_embed_mode.lower()
        if self.spkr_embed_mode == 'adain':                                               # ref: src/module.py:111-122
            self.pseudo_latent_mean = nn.Linear(spkr_embed_dim, query_rnn_dim)
            self.pseudo_latent_std = nn.Sequential(nn.Linear(spkr_embed_dim, query_rnn_dim), nn.ReLU())
        elif self.spkr_embed_mode == 'concat':
            self.spkr_mem_proj = nn.Linear(spkr_embed_dim + enc_embed_dim, enc_embed_dim)
        elif self.spkr_embed_mode == 'add':
            self.spkr_proj = nn.Linear(spkr_embed_dim, enc_embed_dim)
            self.spkr_mem_proj = nn.Linear(enc_embed_dim, enc_embed_dim)
        else:
            raise NotImplementedError
        self.prenet = Prenet(n_mels * n_frames_per_step, [prenet_dim, prenet_dim], apply_dropout=prenet_dropout,
                             norm_type=prenet_norm_type)
        self.query_rnn = nn.LSTMCell(prenet_dim + enc_embed_dim, query_rnn_dim)
        self.attn = Attention(query_rnn_dim, enc_embed_dim, attn_dim, n_location_filters, location_kernel_size,
                              loc_aware, use_summed_weights)
        self.dec_rnn = nn.LSTMCell(query_rnn_dim + enc_embed_dim, dec_rnn_dim)
        self.proj = Linear(dec_rnn_dim + enc_embed_dim, n_mels * n_frames_per_step)
        self.gate_layer = Linear(dec_rnn_dim + enc_embed_dim, 1, bias=True, w_init_gain='sigmoid')
        self.last_tapes = None     # tapes of the most recent forward (saved tensors of the backward pass)
        self.fuse_prenet = True    # inference: emit prenet layer 1 from the proj/gate launch (fp32 re-association)
        self.cache_packed = False  # frozen-weight inference: keep the packed weights across forwards
        # the location conv + W_l part of the attention of step t+1 runs inside the proj launch of step t (st_decoder_io fields)
        self.attn_split = True
        self.attn_pre_parts = 4      # workgroups per utterance of the pre part (measured at L = 43: 1 / 2 / 4 parts 37.1 / 35.9 / 35.4 us per step)
        self.attn_fin_parts = 2      # workgroups per utterance of the fin part (slices of the context dims)
        self.attn_pq_in_fin = True   # inference: query projection and fin part share one launch (in-launch hand-off of pq)
        # ... and the part of the decoder cell's gate product that does not wait for the attention rides in that launch (round 6: +4 %
        # mel-frames/s at C2; fp32 re-association only; ST_SPLIT_GATES=0 restores the whole product in the cell launch)
        self.split_gates = os.environ.get('ST_SPLIT_GATES', '1') != '0'
        self.split_cell_k = int(os.environ.get('ST_SPLIT_CELL_K', '0'))       # reduction columns the decoder cell keeps (0 = st_decoder_gate_split_k)
        self.split_gates_train = os.environ.get('ST_SPLIT_GATES_TRAIN', '1') != '0'      # ... the same split in the teacher-forced training loop
        # the hand-off's failure word (st_decoder_io.handoff_status): an eager forward reads it back right away (one small
        # device -> host copy); under stream capture nobody can, so GraphedDecoder / bench.py / gen_specgram check it after replays
        self.check_handoff = True
        self.handoff_status = None
        self.prenet2_in_proj = os.environ.get('ST_P2', '0') == '1'       # prenet layer 2 inside the proj launch of a free-running step: MEASURED SLOWER (+0.4 us per step), off unless ST_P2=1
        self.attn_split_min_len = 128     # texts at least this long: fin part over position ranges (~attn_split_positions each) + combine
        self.attn_split_positions = 43
        self.bwd_fuse_pointwise = True   # training (teacher forcing): the cells' pointwise backward in the epilogues of the loop's products
        self.bwd_overlap_attn = True     # ... and the decoder cell's product of step t-1 beside the attention backward of step t (one launch)
        self.bwd_dxq_splits = int(os.environ.get('ST_DXQ_SPLITS', '4'))   # ... and the query cell's product likewise (0 = whole)
        self.bwd_dxd_splits = int(os.environ.get('ST_DXD_SPLITS', '2'))   # ... with the decoder cell's product of that launch K-split into slabs (0 = whole)
        self.bwd_attn_parts = int(os.environ.get('ST_ATTN_PARTS', '2'))          # ... that attention backward as 2 workgroups per utterance over halves of the attention dims (1 = whole)
        self.fwd_pair_cells = True       # teacher-forced forward: the decoder cell of step t and the query cell of step t+1 in one launch
        self.attn_rng_one_launch = True   # long texts: query projection + fin part over position ranges + combine in one launch

    # -- helpers ---------------------------------------------------------------------------------
    def gate_split_k(self):
        """reduction columns of the decoder cell's gate product that stay in the cell launch when the rest rides beside pq / fin (split_gates):
        the library's rule for these dimensions unless ST_SPLIT_CELL_K overrides it"""
        if self.split_cell_k:
            return int(self.split_cell_k)
        import ctypes as C
        from . import _lib
        d = StDecoderDims(B=32, L=1, E=self.enc_embed_dim, n_mels=self.n_mels, r=self.n_frames_per_step, P=self.prenet_dim,
                          Q=self.query_rnn_dim, D=self.dec_rnn_dim, A=1)
        return int(_lib.load().st_decoder_gate_split_k(C.byref(d)))

    def _weights_struct(self, keep, fuse_pre0=False):
        w = StDecoderWeights()
        # [proj ; gate (; W_pre0 . W_proj)] assembled with library kernels only, so that a hipGraph capture
        # of this forward re-creates it on replay (torch ops issued during a capture are NOT captured)
        pw, gw = self.proj.linear.weight, self.gate_layer.linear.weight
        in_dim, KO = pw.shape
        n_rows = in_dim + 1 + (self.prenet_dim if fuse_pre0 else 0)
        if not fuse_pre0 and not ops.capturing():
            # (training / eager teacher forcing: the two concatenations are cached parameter layouts, refreshed with every other layout
            # of the model by the one relayout launch of the step -- four copy launches less per forward)
            pg_w = ops.cat_params([pw.detach(), gw.detach()])
            pg_b = ops.cat_params([self.proj.linear.bias.detach(), self.gate_layer.linear.bias.detach()])
        else:
            pg_w = torch.empty(n_rows, KO, device=pw.device, dtype=torch.float32)
            pg_b = torch.empty(n_rows, device=pw.device, dtype=torch.float32)
            ops.copy2d(pg_w, pw, in_dim, KO)
            ops.copy2d(pg_w[in_dim:], gw, 1, KO)
            ops.copy2d(pg_b.view(1, -1), self.proj.linear.bias.view(1, -1), 1, in_dim)
            ops.copy2d(pg_b.view(1, -1)[:, in_dim:], self.gate_layer.linear.bias.view(1, -1), 1, 1)
        if fuse_pre0:
            # prenet layer 1 of the own output, folded into the projection: relu(W0 (Wp y + bp)) =
            # relu((W0 Wp) y + W0 bp); the two products are formed once per forward on device
            w0 = self.prenet.layers[0].linear.weight
            pwT = pw.detach().t().contiguous()            # parameter layout change only (no captured input)
            ops.gemm(w0, pwT, out=pg_w[in_dim + 1:])
            ops.gemm(w0, self.proj.linear.bias.view(1, -1), out=pg_b[in_dim + 1:].view(-1, 1))
            keep.append(pwT)
        loc_wc, loc_wl = (t.detach().contiguous() for t in self.attn.location_weights())
        keep += [loc_wc, loc_wl]
        keep += [pg_w, pg_b]           # (the last two entries: the deferred projection of _run_loop reads them)
        tensors = dict(
            prenet_w0=self.prenet.layers[0].linear.weight, prenet_w1=self.prenet.layers[1].linear.weight,
            q_w_ih=self.query_rnn.weight_ih, q_w_hh=self.query_rnn.weight_hh,
            q_b_ih=self.query_rnn.bias_ih, q_b_hh=self.query_rnn.bias_hh,
            attn_query_w=self.attn.query_layer.linear.weight, attn_v=self.attn.v.linear.weight,
            attn_loc_conv_w=loc_wc, attn_loc_lin_w=loc_wl,
            d_w_ih=self.dec_rnn.weight_ih, d_w_hh=self.dec_rnn.weight_hh,
            d_b_ih=self.dec_rnn.bias_ih, d_b_hh=self.dec_rnn.bias_hh, projgate_w=pg_w, projgate_b=pg_b)
        for k, t in tensors.items():
            setattr(w, k, ops._p(t))
        if self.prenet_norm_type is not None:
            for l, layer in enumerate(self.prenet.layers):
                nm = layer.norm
                w.pre_norm_w[l], w.pre_norm_b[l] = ops._p(nm.weight), ops._p(nm.bias)
                if self.prenet_norm_type == 'BatchNorm1d':
                    w.pre_norm_rm[l], w.pre_norm_rv[l] = ops._p(nm.running_mean), ops._p(nm.running_var)
                    w.pre_norm_nbt[l] = ops._p(nm.num_batches_tracked, torch.int64)
                    w.pre_norm_momentum = float(nm.momentum)
                w.pre_norm_eps = float(nm.eps)
        return w

    def forward(self, memory, memory_lengths, teacher, spkr_embed, tf_rate=0.0, unpair_max_frame=None, _masks=None):
        """Same contract as the reference (src/module.py:140-214):
        memory (B,L,E); teacher int (max frames, inference) or (Bt,T,n_mels); spkr_embed (B,S).
        Returns mel (B, steps*r, n_mels), alignments (B, steps, L), stops (B, steps*r).
        `_masks` (tests only): dict with explicit scaled dropout masks 'teacher' [m1, m2],
        'own' (steps,2,B,P), 'q' (steps,B,Q), 'd' (steps,B,D)."""
        assert len(self.prenet.layers) == 2
        dev = memory.device
        memory = memory.contiguous()
        spkr_embed = spkr_embed.contiguous()
        B, L, E = memory.shape
        r, n_mels, P = self.n_frames_per_step, self.n_mels, self.prenet_dim
        Q, D, A = self.query_rnn_dim, self.dec_rnn_dim, self.attn_dim
        _masks = _masks or {}
        is_int = isinstance(teacher, int)
        Bt = B if is_int else teacher.shape[0]
        steps, step_src = plan_decode(is_int, teacher if is_int else teacher.shape[1], Bt, B, r, tf_rate,
                                      self.drop_dec_in, unpair_max_frame)
        differentiable = self.training and torch.is_grad_enabled()
        lin = (lambda x, m, act=None: AG.linear(x, m.weight, m.bias, act)) if differentiable else \
              (lambda x, m, act=None: (ops.linear_small(x, m.weight, m.bias, act) if x.dim() == 2 and x.shape[0] <= 64
                                       else ops.gemm(x, m.weight, bias=m.bias, act_pre=act)))
        # once per utterance (everything the reference recomputes per step from step-invariant inputs is hoisted out of the loop):
        # processed memory; AdaIN statistics, or -- spkr_embed_mode 'concat' / 'add' -- the speaker-conditioned memory the context
        # is read from (src/module.py:243-250; the processed memory still comes from the plain memory, :306), with the AdaIN of the
        # query state reduced to the identity (std 1, mean 0, :271-272)
        pm = AG.conv(memory, self.attn.memory_layer.linear.weight) if differentiable else self.attn.process_memory(memory)
        ctx_memory = memory
        if self.spkr_embed_mode == 'adain':
            ada_std = lin(spkr_embed, self.pseudo_latent_std[0], 'relu')
            ada_mean = lin(spkr_embed, self.pseudo_latent_mean)
        else:
            ada_std = torch.ones(B, Q, device=dev, dtype=torch.float32)
            ada_mean = torch.zeros(B, Q, device=dev, dtype=torch.float32)
            if self.spkr_embed_mode == 'concat':
                # Linear(cat[mem, spkr]) = W[:, :E] mem + (W[:, E:] spkr + b): the speaker part once per utterance, added per position
                w = self.spkr_mem_proj.weight
                sp = (AG.linear(spkr_embed, w[:, E:].contiguous(), self.spkr_mem_proj.bias) if differentiable
                      else ops.linear_small(spkr_embed, w[:, E:].contiguous(), self.spkr_mem_proj.bias))
                res = sp.unsqueeze(1).expand(B, L, E).contiguous()
                ctx_memory = (AG.conv(memory, w[:, :E].contiguous(), res=res) if differentiable
                              else ops.gemm(memory, w[:, :E].contiguous(), res=res))
            else:                                                                          # 'add'
                sp = lin(spkr_embed, self.spkr_proj)
                ctx_memory = lin(memory + sp.unsqueeze(1), self.spkr_mem_proj)
        if self.pretrain:
            # decoder pre-training without attention (:238-240): context and alignments are zero.  A zero memory gives the zero
            # context through the same kernels; the alignments handed back are zeroed below
            ctx_memory = torch.zeros_like(memory)
        teacher_pre = teacher_mean = None
        Tt = 0
        if tf_rate != 0.0:
            tch = teacher.contiguous().view(Bt, -1, n_mels * r)                                   # :178
            Tt = tch.shape[1]
            teacher_pre = self.prenet(tch, _masks.get('teacher'))                                 # :179
            if any(s == -2 for s in step_src):
                teacher_mean = ops.mean_rows(teacher_pre.detach())
        # the go frame: prenet(0) is 0 for a plain prenet (bias-free Linear + ReLU), so the loop starts from a zero input; a
        # normalised prenet maps the zero frame to relu(beta) (LayerNorm) / relu((0 - mean) / sigma * gamma + beta) (BatchNorm1d)  :161,:183
        dec_in0 = None
        if self.prenet_norm_type is not None:
            dec_in0 = self.prenet(torch.zeros(B, n_mels * r, device=dev, dtype=torch.float32), _masks.get('go')).contiguous()
        uses_own = any(s == -1 for s in step_src[:-1]) or Bt < B
        own_mask = _masks.get('own')
        if own_mask is None and uses_own and self.prenet_dropout > 0:
            own_mask = _scaled_mask((steps, 2, B, P), self.prenet_dropout, dev)
        q_mask, d_mask = _masks.get('q'), _masks.get('d')
        if self.training:
            if q_mask is None and d_mask is None and self.query_dropout.p == self.dec_dropout.p and self.query_dropout.p > 0:
                q_mask, d_mask = _scaled_masks([(steps, B, Q), (steps, B, D)], self.query_dropout.p, dev)      # (one draw for both cells)
            if q_mask is None and self.query_dropout.p > 0:
                q_mask = _scaled_mask((steps, B, Q), self.query_dropout.p, dev)
            if d_mask is None and self.dec_dropout.p > 0:
                d_mask = _scaled_mask((steps, B, D), self.dec_dropout.p, dev)
        plan = dict(steps=steps, step_src=step_src, Bt=Bt, Tt=Tt, masks=(own_mask, q_mask, d_mask),
                    teacher_mean=teacher_mean, dec_in0=dec_in0.detach() if dec_in0 is not None else None)
        if differentiable:
            mel, align, stop = AG.decoder_loop(self, plan, ctx_memory.contiguous(), pm, ada_std, ada_mean, teacher_pre, dec_in0)
            return mel, (align * 0.0 if self.pretrain else align), stop
        mel, align, stop, tapes = self._run_loop(plan, ctx_memory.contiguous(), pm, ada_std, ada_mean, teacher_pre, keep_tapes=False)
        self.last_tapes = tapes
        if self.pretrain:
            ops.fill_(align, 0.0)
        return mel, align, stop

    def _run_loop(self, plan, memory, pm, ada_std, ada_mean, teacher_pre, keep_tapes):
        """the decode loop itself (st_decoder_forward) on plain tensors; `keep_tapes` also records what the
        backward needs (activated LSTM gates)"""
        dev = memory.device
        B, L, E = memory.shape
        r, n_mels, P = self.n_frames_per_step, self.n_mels, self.prenet_dim
        Q, D, A = self.query_rnn_dim, self.dec_rnn_dim, self.attn_dim
        steps, step_src, Bt, Tt = plan['steps'], plan['step_src'], plan['Bt'], plan['Tt']
        own_mask, q_mask, d_mask = plan['masks']
        teacher_mean = plan['teacher_mean']
        keep = []
        f32 = dict(device=dev, dtype=torch.float32)
        mel = torch.empty(B, steps * r, n_mels, **f32)
        align = torch.empty(B, steps, L, **f32)
        stop = torch.empty(B, steps * r, **f32)
        lib = _lib.load()
        # free-running inference: fold prenet layer 1 into the proj/gate launch (one kernel less per step)
        fuse_pre0 = ((not self.training) and self.fuse_prenet and Bt == B and all(s_ == -1 for s_ in step_src)
                     and self.prenet_norm_type is None)
        # normalised prenet: 1 LayerNorm, 2 BatchNorm1d with running statistics (eval), 3 BatchNorm1d with a step's batch statistics
        pn_mode = {None: 0, 'LayerNorm': 1, 'BatchNorm1d': 3 if self.training else 2}[self.prenet_norm_type]
        dims = StDecoderDims(B=B, L=L, E=E, n_mels=n_mels, r=r, P=P, Q=Q, D=D, A=A,
                             F=self.n_location_filters, K=self.location_kernel_size, fuse_pre0=1 if fuse_pre0 else 0,
                             prenet_norm=pn_mode)
        t16 = lambda k: int(lib.st_t16_floats(B, k))           # floats of one T16-tiled (B, k) buffer
        slot = [int(lib.st_decoder_tape_floats(C.byref(dims), i)) for i in range(3)]
        in_dim = r * n_mels
        # T16-tiled step-input tapes (xq = [dec_in|ctx|h_q], xd = [ctx|adapted h_q|h_d], xo = [h_d|ctx]) are
        # handed in zero-filled (pad lanes/rows must be zero); ONE memset covers them all
        # training keeps the prenet layer-1 output of every step (own-output feedback needs it in the backward)
        n_pre1 = steps if keep_tapes else 1
        sizes = dict(xq=(steps + 1) * slot[0], xd=(steps + 1) * slot[1], xo=steps * slot[2], pre1=n_pre1 * t16(P), melt=t16(in_dim))
        # ... unless nothing is padded (B and every width a multiple of 16) and every element is written before it is read: pure teacher
        # forcing in training (all dec_in slots by the teacher tiling, ctx / h parts by the steps, slot 0 by the loop's zero launch;
        # the own-output scratch is never touched) -- the C2 training step saves an 11 us fill of 90 MB
        no_pads = B % 16 == 0 and all(k % 16 == 0 for k in (P, E, Q, D, in_dim))
        tf_all = teacher_pre is not None and Bt == B and all(step_src[t] == min(t, Tt - 1) for t in range(steps - 1))
        tiled = (ops.uninit if (keep_tapes and no_pads and tf_all and self.prenet_norm_type is None) else ops.zeros)(sum(sizes.values()), **f32)
        tapes, off = {}, 0
        for k, n in sizes.items():
            tapes[k] = tiled[off:off + n]
            off += n
        tapes.update(cq=torch.empty(steps + 1, B, Q, **f32), cd=torch.empty(steps + 1, B, D, **f32),
                     wcum=torch.empty(steps + 1, B, L, **f32), pq=torch.empty(B, A, **f32),
                     zero=torch.empty(B, L, **f32), tiled=tiled)
        if keep_tapes:
            tapes['gates_q'] = torch.empty(steps, B, 4, Q, **f32)
            tapes['gates_d'] = torch.empty(steps, B, 4, D, **f32)

        # the six matrices the loop streams every step, packed into MFMA lane order: once per forward while the weights
        # can change (training), once per weight version for frozen-weight inference (`cache_packed`, see
        # runtime.GraphedDecoder.refresh_weights)
        cache = self.__dict__.setdefault('_packed_cache', {}) if (self.cache_packed and not self.training) else None
        # keyed by the in-place version counters of the decoder's parameters: load_state_dict / an optimiser step bump them,
        # so a stale packing is never reused by an eager forward (a captured graph still needs refresh_weights())
        ckey = (bool(fuse_pre0), str(dev), tuple(p._version for p in self.parameters())) if cache is not None else None
        if cache is not None and ckey not in cache:
            cache.clear()
        if cache is not None and ckey in cache:
            w, keep_w, packed = cache[ckey]
            keep.append(keep_w)
        else:
            w = self._weights_struct(keep, fuse_pre0)
            packed = torch.empty(int(lib.st_decoder_packed_floats(C.byref(dims))), **f32)
            check(lib.st_decoder_pack(C.byref(w), C.byref(dims), ops._p(packed), ops.stream_handle()), 'st_decoder_pack')
            if cache is not None:
                cache[ckey] = (w, list(keep), packed)
        io = StDecoderIO()
        src_arr = (C.c_int * max(steps, 1))(*step_src)
        io.memory, io.pm, io.ada_std, io.ada_mean = ops._p(memory), ops._p(pm), ops._p(ada_std), ops._p(ada_mean)
        io.step_src = C.cast(src_arr, C.POINTER(C.c_int))
        io.teacher_pre, io.teacher_mean = ops._p(teacher_pre), ops._p(teacher_mean)
        io.Bt, io.Tt = Bt, Tt
        io.prenet_mask, io.q_mask, io.d_mask = ops._p(own_mask), ops._p(q_mask), ops._p(d_mask)
        io.steps = steps
        io.mel_out, io.align_out, io.stop_out = ops._p(mel), ops._p(align), ops._p(stop)
        io.packed = ops._p(packed)
        io.xq_tape, io.xd_tape, io.xo_tape = (ops._p(tapes[k]) for k in ('xq', 'xd', 'xo'))
        io.cq_tape, io.cd_tape, io.wcum_tape = (ops._p(tapes[k]) for k in ('cq', 'cd', 'wcum'))
        io.pq_buf, io.pre1_t16, io.mel_t16 = (ops._p(tapes[k]) for k in ('pq', 'pre1', 'melt'))
        io.zero_row = ops._p(tapes['zero'])
        if pn_mode:
            tapes['pre_nat'] = torch.empty(B, P, **f32)
            io.pre_nat = ops._p(tapes['pre_nat'])
            tapes['pn_mode'] = pn_mode
            if keep_tapes:      # the Linear outputs of both layers of every own-output feedback: the norm backward starts from them
                tapes['pre_y'] = torch.empty(steps, 2, B, P, **f32)
                io.pre_nat_tape = ops._p(tapes['pre_y'])
        if plan.get('dec_in0') is not None:
            io.dec_in0 = ops._p(plan['dec_in0'])
        io.gates_q_tape, io.gates_d_tape = ops._p(tapes.get('gates_q')), ops._p(tapes.get('gates_d'))
        io.pre1_step_floats = t16(P) if keep_tapes else 0
        # training with pure teacher forcing: no step's input depends on an earlier output, so mel / stop of all steps
        # come from ONE GEMM over the xo tape after the loop instead of one launch per step
        pure_tf = teacher_pre is not None and Bt == B and all(step_src[t] == min(t, Tt - 1) for t in range(steps - 1))
        defer = bool(keep_tapes and pure_tf)
        io.defer_proj = 1 if defer else 0
        io.pair_cells = 1 if (defer and self.fwd_pair_cells) else 0
        if self.attn_split and (not self.training or defer):
            if keep_tapes:      # training: S and the location features of every step stay for the backward pass
                tapes['attn_s'] = ops.uninit(steps, B, L, A, **f32)
                tapes['attn_loc'] = ops.uninit(steps, B, L, self.n_location_filters, **f32)     # (slot 0 -- no history yet -- is zeroed by the loop's first launch)
                io.attn_s_step_floats = B * L * A
                io.attn_loc_tape = ops._p(tapes['attn_loc'])
            else:
                tapes['attn_s'] = torch.empty(B, L, A, **f32)
            io.attn_s_buf = ops._p(tapes['attn_s'])
            parts = int(self.attn_pre_parts)
            while parts < 64 and (L + parts - 1) // parts > 512:      # long texts: a workgroup holds the conv features of its
                parts *= 2                                            # position range in LDS (32 filters x <= ~512 positions)
            io.attn_pre_parts = parts
            io.attn_fin_parts = int(self.attn_fin_parts)
            if L >= self.attn_split_min_len and not keep_tapes:
                # long texts: fin part split over position ranges + a combine launch (one CU cannot pull 2 L A 4 bytes fast enough)
                sp = max(2, min(64, (L + self.attn_split_positions - 1) // self.attn_split_positions))
                tapes['attn_split_ws'] = torch.empty(int(lib.st_attn_fin_split_workspace_floats(B, E, sp)), **f32)
                io.attn_split_ws, io.attn_split_parts = ops._p(tapes['attn_split_ws']), sp
                if self.attn_pq_in_fin and sp <= 8 and self.attn_rng_one_launch:
                    # ... and, when the device holds every workgroup at once, as ONE launch with the combine inside (the library
                    # decides: st_query_attn_rng_fits; the three-launch form stays the fall-back)
                    tapes['attn_xchg'] = torch.empty(2 * int(lib.st_attn_rng_xchg_words(B, E, sp)), **f32)
                    io.attn_xchg = ops._p(tapes['attn_xchg'])
            if self.split_gates and 16 < B <= 32 and ((not keep_tapes and L < self.attn_split_min_len) or (defer and self.split_gates_train)):
                # the tail of the decoder cell's gate products -- columns of [ctx | AdaIN(h_q(t)) | h_d(t-1)] known before the attention runs --
                # beside the pq / fin launch (st_decoder_io.gate_part; in the other forms of the attention step a launch of its own: the same
                # arithmetic whatever the form).  gate_part_k = 0: the library's split (half of the reduction); ST_SPLIT_CELL_K for ablations.
                # Teacher-forced training (defer): the product rides beside the query projection + attention pre part, the paired cell launch adds it
                tapes['gate_part'] = torch.empty(B, 4 * D, **f32)
                io.gate_part = ops._p(tapes['gate_part'])
                io.gate_part_k = int(self.split_cell_k)
            if self.attn_pq_in_fin and not keep_tapes:
                # query projection + attention fin part as ONE launch per step (st_query_attn_fin_fwd): pq is handed over inside
                # the launch as 8-byte {value, tag} words; the library falls back to two launches when the shapes do not fit
                tapes['pq_gran'] = torch.empty(B, 2 * A, **f32)            # (B, A) 64-bit words
                io.pq_granules = ops._p(tapes['pq_gran'])
                if self.handoff_status is None or self.handoff_status.device != dev:
                    self.handoff_status = torch.zeros(1, device=dev, dtype=torch.int32)
                io.handoff_status = ops._p(self.handoff_status, torch.int32)
                if fuse_pre0 and self.prenet2_in_proj:
                    # ... and prenet layer 2 of the next input inside the proj (+) gate (+) prenet-layer-1 launch (its operand as granules):
                    # three launches per free-running decode step + the two cells, instead of four
                    tapes['pre1_gran'] = torch.empty(B, 2 * P, **f32)       # (B, P) 64-bit words
                    io.pre1_granules = ops._p(tapes['pre1_gran'])
        check(lib.st_decoder_forward(C.byref(w), C.byref(dims), C.byref(io), ops.stream_handle()),
              'st_decoder_forward')
        if io.handoff_status and self.check_handoff and not ops.capturing():
            if ops.handoff_starved(self.handoff_status):
                # the fin workgroups ran without their producers (a shared / CU-masked GPU): the pass is NaN.  Run it again as two
                # launches per hand-off -- same tapes (every slot is rewritten, the initial slots are re-zeroed by the loop's first
                # launch), same masks -- and keep that form for the rest of the process
                ops.degrade('decode loop (query projection -> attention hand-off)', 'one launch each (attn_pq_in_fin = False)')
                self.attn_pq_in_fin = False
                io.pq_granules, io.handoff_status, io.attn_xchg, io.pre1_granules = None, None, None, None
                check(lib.st_decoder_forward(C.byref(w), C.byref(dims), C.byref(io), ops.stream_handle()), 'st_decoder_forward')
        if not keep_tapes and self.check_handoff and not ops.capturing():
            ops.check_persist_status(dev)           # (the text encoder's one-launch BiLSTM ran before this loop: Tacotron2.forward falls back)
        if defer:
            kb = ops.kb16
            Bp = ((B + 15) // 16) * 16
            xo_nat = ops.untile_tape(tapes['xo'], steps, Bp, kb(D) + kb(E), [(0, D), (kb(D), E)])       # [h_d_t | ctx_t]
            y = ops.gemm(xo_nat.view(-1, D + E), keep[-2], bias=keep[-1]).view(steps, Bp, in_dim + 1)   # proj (+) gate rows
            # rows (t, b) of [mel_t | stop_t] -> mel (B, steps*r, n_mels), stop.repeat(1, r) (:287): one launch
            check(lib.st_decoder_unpack_out(ops._p(y), ops._p(mel), ops._p(stop), B, Bp, steps, r, n_mels, in_dim + 1, ops.stream_handle()),
                  'st_decoder_unpack_out')
            tapes['xo_nat'] = xo_nat
        tapes['packed'] = packed
        tapes.update(pm=pm, ada_std=ada_std, ada_mean=ada_mean, teacher_pre=teacher_pre, masks=(own_mask, q_mask, d_mask),
                     keep=keep, step_src=step_src, slot=slot)
        return mel, align, stop, tapes


# ----------------------------------------------------------------------------- CBHG
class BatchNormConv1d(nn.Module):
    """conv1d(no bias) -> activation -> BatchNorm1d(momentum .99, eps 1e-3).  ref: src/module.py:527-538"""

    def __init__(self, in_size, out_size, kernel_size, stride, padding, activation=None):
        super().__init__()
        assert stride == 1
        self.conv1d = nn.Conv1d(in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm1d(out_size, momentum=0.99, eps=1e-3)
        self.activation = activation
        self.padding = padding

    def forward(self, x_cl, training=None, Tout=None, out=None, coff=0, pool_prev=False, stats_Tout=None, collect=None):
        order = 'relu_bn' if self.activation is not None else 'bn'
        return _conv_bn_act(x_cl, self.conv1d.weight, None, self.bn, self.padding, self.bn.eps, self.bn.momentum,
                            order, self.training if training is None else training, Tout, out, coff, pool_prev,
                            stats_Tout, collect)


class Highway(nn.Module):
    """y = relu(H x) * sigmoid(T x) + x * (1 - sigmoid(T x)).  ref: src/module.py:541-555"""

    def __init__(self, in_size, out_size):
        super().__init__()
        self.H = nn.Linear(in_size, out_size)
        self.H.bias.data.zero_()
        self.T = nn.Linear(in_size, out_size)
        self.T.bias.data.fill_(-1)

    def forward(self, x):
        if self.training:
            return AG.highway_layer(x, self.H.weight, self.H.bias, self.T.weight, self.T.bias)
        h = ops.gemm(x, self.H.weight, bias=self.H.bias, act_pre='relu')
        return ops.gemm(x, self.T.weight, bias=self.T.bias, act_pre='sigmoid', highway_h=h, res=x)


class CBHG(nn.Module):
    """conv bank (k = 1..K) -> max-pool -> conv projections -> highways -> BiGRU.
    ref: src/module.py:558-622"""

    def __init__(self, in_dim, K=16, hidden_sizes=(128, 128)):
        super().__init__()
        hidden_sizes = list(hidden_sizes)
        self.in_dim = in_dim
        relu = nn.ReLU()
        self.conv1d_banks = nn.ModuleList([BatchNormConv1d(in_dim, in_dim, kernel_size=k, stride=1, padding=k // 2,
                                                           activation=relu) for k in range(1, K + 1)])
        in_sizes = [K * in_dim] + hidden_sizes[:-1]
        acts = [relu] * (len(hidden_sizes) - 1) + [None]
        self.conv1d_projs = nn.ModuleList([BatchNormConv1d(i, o, kernel_size=3, stride=1, padding=1, activation=a)
                                           for i, o, a in zip(in_sizes, hidden_sizes, acts)])
        self.pre_highway_proj = nn.Linear(hidden_sizes[-1], in_dim, bias=False)
        self.highways = nn.ModuleList([Highway(in_dim, in_dim) for _ in range(4)])
        self.gru = nn.GRU(in_dim, in_dim, num_layers=1, batch_first=True, bidirectional=True)

    def forward(self, inputs, input_lengths=None):
        assert input_lengths is None, 'the reference never passes lengths to the postnet CBHG'
        x = inputs.contiguous()
        B, T, Cn = x.shape
        assert Cn == self.in_dim
        K = len(self.conv1d_banks)
        if self.training:
            return self._forward_train(x)
        bank = torch.empty(B, T, K * Cn, device=x.device, dtype=torch.float32)
        jobs = []        # the K convs read the same input: one launch for all of them where the kernels allow (ops.gemm_flush)
        for i, blk in enumerate(self.conv1d_banks):
            k = i + 1
            # even k yields T+1 positions and, in training mode, BatchNorm sees all of them before
            # the trim to T (module.py:597-598): stats_Tout tells the helper to include the extra one
            blk(x, Tout=T, out=bank, coff=i * Cn, stats_Tout=T + 1 if k % 2 == 0 else T, collect=jobs)
        ops.gemm_flush(jobs)
        y = self.conv1d_projs[0](bank, pool_prev=True)       # MaxPool1d(2,1,1)[:T] fused into the load
        for blk in self.conv1d_projs[1:]:
            y = blk(y)
        y = ops.gemm(y, self.pre_highway_proj.weight, res=x)                                  # :607-609
        ys = ops.highway_stack(y, [(hw.H.weight, hw.H.bias, hw.T.weight, hw.T.bias) for hw in self.highways])      # one launch
        if ys is None:
            for hw in self.highways:
                y = hw(y)
        else:
            y = ys
        H = self.gru.hidden_size
        jobs = []        # the two directions' input projections: one launch
        gi_f = ops.gemm(y, self.gru.weight_ih_l0, bias=self.gru.bias_ih_l0, collect=jobs)
        gi_b = ops.gemm(y, self.gru.weight_ih_l0_reverse, bias=self.gru.bias_ih_l0_reverse, collect=jobs)
        ops.gemm_flush(jobs)
        out = torch.empty(B, T, 2 * H, device=x.device, dtype=torch.float32)
        ops.gru_seq(gi_f, gi_b, self.gru.weight_hh_l0, self.gru.weight_hh_l0_reverse,
                    self.gru.bias_hh_l0, self.gru.bias_hh_l0_reverse, out)
        return out


def _cbhg_forward_train(self, x):
    """differentiable CBHG (same kernels; every stage keeps what its backward needs)"""
    B, T, Cn = x.shape
    # even k yields T+1 positions and BatchNorm sees all of them before the trim to T (module.py:597-598)
    # (the K BatchNorms go through ONE autograd function: under SyncBN they share one all-gather and one all-reduce)
    # (... and the K convolutions through ONE autograd function: one forward launch, the input gradient accumulated by the products)
    acts = set(blk.activation is not None for blk in self.conv1d_banks)
    assert len(acts) == 1
    relu = acts.pop()
    bank_launches = len(self.conv1d_banks) <= 16      # (also under SyncBN: one all-gather and one all-reduce for the whole bank)
    pre = AG.conv_group(x, [blk.conv1d.weight for blk in self.conv1d_banks], [blk.padding for blk in self.conv1d_banks],
                        [T + 1 if (i + 1) % 2 == 0 else T for i in range(len(self.conv1d_banks))], act='relu' if relu else None,
                        act_grad_done=bank_launches and relu)
    if bank_launches:      # the K BatchNorms as three launches, straight into the concatenated bank
        bank = AG.batch_norm_bank(pre, [blk.bn for blk in self.conv1d_banks], T, relu_in=relu)
    else:
        bank = torch.cat([y[:, :T] for y in AG.batch_norm_train_group(pre, [blk.bn for blk in self.conv1d_banks])], dim=-1)
    y = self.conv1d_projs[0](bank, pool_prev=True)
    for blk in self.conv1d_projs[1:]:
        y = blk(y)
    y = AG.conv(y, self.pre_highway_proj.weight, res=x)                                       # :607-609
    for hw in self.highways:
        y = hw(y)
    g = self.gru
    gi_f, gi_b = AG.conv_group(y, [g.weight_ih_l0, g.weight_ih_l0_reverse], [0, 0], [None, None],
                               biases=[g.bias_ih_l0, g.bias_ih_l0_reverse])         # one launch; dy accumulated by the products
    return AG.bigru(gi_f, gi_b, g.weight_hh_l0, g.bias_hh_l0, g.weight_hh_l0_reverse, g.bias_hh_l0_reverse)


CBHG._forward_train = _cbhg_forward_train


class Postnet(nn.Module):
    """The Tacotron-2 5-conv postnet class: defined and imported by the reference but never
    constructed by any config.  ref: src/module.py:53-82"""

    def __init__(self, n_mels, postnet_embed_dim, postnet_kernel_size, postnet_n_conv, postnet_dropout):
        super().__init__()
        in_size = [n_mels] + [postnet_embed_dim] * (postnet_n_conv - 1)
        out_size = [postnet_embed_dim] * (postnet_n_conv - 1) + [n_mels]
        act_fn = ['tanh'] * (postnet_n_conv - 1) + ['linear']
        self.postnet_dropout = postnet_dropout
        self.convs = nn.ModuleList([
            nn.Sequential(Conv1d(din, dout, kernel_size=postnet_kernel_size, stride=1,
                                 padding=(postnet_kernel_size - 1) // 2, dilation=1, w_init_gain=fn),
                          nn.BatchNorm1d(dout), nn.Tanh() if fn == 'tanh' else nn.Identity(), nn.Dropout(postnet_dropout))
            for din, dout, fn in zip(in_size, out_size, act_fn)])

    def forward(self, x, _masks=None):
        """_masks (tests): one scaled dropout mask (B, T, C) per block, replayed instead of drawn"""
        x = x.contiguous()
        for i, blk in enumerate(self.convs):
            conv, bn = blk[0], blk[1]
            order = 'bn_tanh' if isinstance(blk[2], nn.Tanh) else 'bn'
            x = _conv_bn_act(x, conv.conv.weight, conv.conv.bias, bn, conv.padding, bn.eps, bn.momentum, order,
                             self.training)
            if self.training and self.postnet_dropout > 0:                                   # nn.Dropout of the block, :73
                x = x * (_masks[i] if _masks is not None else _scaled_mask(x.shape, self.postnet_dropout, x.device))
        return x
