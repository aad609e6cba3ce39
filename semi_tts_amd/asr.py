"""CTC speech encoder on the HIP path: mirror of the reference's src/asr.py:5-64 and
`ConvLayer` (src/module.py:627-648) -- same constructor arguments, attributes (`time_reduce_factor`, `out_dim`)
and state_dict keys (`layer{l}.conv.*`, `layer{l}.bn.*`, `rnn.*`, `postnet.*`).

Every ConvLayer is ONE launch of the implicit-GEMM conv (stride in the row mapping) with bias -> BatchNorm ->
activation -> residual add -> dropout mask fused in its epilogue; the BiLSTM layers reuse the encoder's
sequence kernels.  In training mode with gradients enabled the same kernels run as differentiable pieces
(semi_tts_amd/autograd.py: conv incl. the stride-2 layer, batch-statistics BatchNorm, BiLSTM), so the CTC half of
the reference's training step (bin/train_vqvae.py:208-217,270) reaches the speech encoder.
"""
import torch
import torch.nn as nn

from . import ops
from . import autograd as AG

_ACT = {'tanh': 'tanh', 'relu': 'relu', 'sigmoid': 'sigmoid'}


class ConvLayer(nn.Module):
    def __init__(self, in_dim, out_dim, kernel_size, stride, residual, batch_norm, activation, dropout):
        super().__init__()
        self.residual, self.batch_norm, self.stride = bool(residual), batch_norm, stride
        self.activation = _ACT[activation.lower()]
        self.padding = 1 if kernel_size != 1 else 0
        self.conv = nn.Conv1d(in_dim, out_dim, kernel_size, stride, padding=self.padding)
        if batch_norm:
            self.bn = nn.BatchNorm1d(out_dim)
        self.drop = nn.Dropout(dropout)

    def out_len(self, T):
        return (T + 2 * self.padding - self.conv.kernel_size[0]) // self.stride + 1

    def forward(self, x, mask=None):
        """x (B,T,C) channels-last (the reference keeps (B,C,T)); `mask`: the scaled dropout mask of the output (None: drawn here)"""
        w, b = self.conv.weight, self.conv.bias
        p = self.drop.p if self.training else 0.0
        if p > 0 and mask is None:
            from .module import _scaled_mask
            mask = _scaled_mask((x.shape[0], self.out_len(x.shape[1]), w.shape[0]), p, x.device)
        res = x if self.residual else None
        if self.training and self.batch_norm and w.shape[0] % 4 == 0:
            if torch.is_grad_enabled():
                # ONE autograd node: conv (+ bias) -> BatchNorm over the batch -> activation -> + x -> dropout      (src/module.py:638-648)
                return AG.conv_layer(x, self.conv, self.bn, mask, self.padding, self.stride, self.activation, self.residual)
            # no gradients: the same launches without the kept tensors (batch statistics of the biased conv output, then the fused tail)
            bn = self.bn
            y = ops.gemm(x.contiguous(), w, pad=self.padding, stride=self.stride, bias=b)
            y2 = y.view(-1, y.shape[-1])
            mean, var = ops.bn_stats(y2, 0, y.shape[-1], bn.running_mean, bn.running_var, bn.momentum, bn.num_batches_tracked)
            _, out = ops.bn_norm_res_mask(y2, mean, var, bn.weight, bn.bias, bn.eps, self.activation,
                                          res.contiguous().view(-1, y.shape[-1]) if res is not None else None,
                                          mask.view(-1, y.shape[-1]) if mask is not None else None, want_t=False)
            return out.view(y.shape)
        if self.training and torch.is_grad_enabled():
            # (layers the fused node does not take: no BatchNorm, or a channel count that is not a multiple of 4)
            if self.batch_norm:
                y = AG.batch_norm_train(AG.conv(x, w, b, pad=self.padding, stride=self.stride), self.bn, self.activation)
            else:
                y = AG.conv(x, w, b, pad=self.padding, stride=self.stride, act=self.activation)
            if self.residual:
                y = y + x
            return AG.mask_mul(y, mask) if mask is not None else y
        if not self.batch_norm:
            return ops.gemm(x, w, pad=self.padding, stride=self.stride, bias=b, act_post=self.activation, res=res, mask=mask)
        bn = self.bn
        if not self.training:
            return ops.gemm(x, w, pad=self.padding, stride=self.stride, bias=b,
                            bn=(bn.running_mean, bn.running_var, bn.weight, bn.bias), bn_eps=bn.eps,
                            act_post=self.activation, res=res, mask=mask)
        # training without gradients, odd channel count: batch statistics of the biased conv output, then the fused pass with them
        y = ops.gemm(x, w, pad=self.padding, stride=self.stride, bias=b)
        mean, var = ops.bn_stats(y.view(-1, y.shape[-1]), 0, y.shape[-1], bn.running_mean, bn.running_var, bn.momentum,
                                 bn.num_batches_tracked)
        return ops.gemm(x, w, pad=self.padding, stride=self.stride, bias=b, bn=(mean, var, bn.weight, bn.bias),
                        bn_eps=bn.eps, act_post=self.activation, res=res, mask=mask)


class CTC(nn.Module):
    def __init__(self, in_dim, out_dim, dim, dropout, kernel, stride, residual, batch_norm, activation,
                 rnn_layers, rnn_dim, rnn_bid, layer_norm):
        super().__init__()
        self.kernel, self.stride, self.residual = kernel, stride, residual
        self.layers = len(kernel)
        self.dim = [in_dim] + ([dim] * self.layers if isinstance(dim, int) else list(dim))
        self.rnn_dim, self.batch_norm, self.layer_norm = rnn_dim, batch_norm, layer_norm
        self.out_dim, self.dropout = out_dim, dropout
        self.time_reduce_factor = 2 ** sum(1 for s in stride if s != 1)                      # :22
        for l in range(self.layers):
            setattr(self, 'layer' + str(l), ConvLayer(self.dim[l], self.dim[l + 1], kernel[l], stride[l], residual[l],
                                                      batch_norm, activation, dropout))
        assert rnn_dim > 0
        self.rnn_bid = bool(rnn_bid)
        self.rnn = nn.LSTM(self.dim[-1], rnn_dim, num_layers=rnn_layers, dropout=dropout, bidirectional=self.rnn_bid, batch_first=True)
        self.rnn_layers = rnn_layers
        cur = rnn_dim * 2 if self.rnn_bid else rnn_dim                                         # :37
        if self.layer_norm:
            self.norm_layer = nn.LayerNorm(cur)                                                # :38-39
        self.drop = nn.Dropout(dropout)
        self.postnet = nn.Linear(cur, out_dim)

    def forward(self, x, _masks=None):
        """x (B,T,n_mels) -> (B, T / time_reduce_factor, out_dim)                            ref: src/asr.py:46-64
        `_masks` (tests only): explicit scaled dropout masks, one per conv layer then one per LSTM layer (None = draw)"""
        x = x.contiguous()
        p = self.dropout if self.training else 0.0
        if _masks is None and p > 0:
            # every dropout mask of the encoder (one per conv layer, one per LSTM layer: the same rate) from ONE draw
            from .module import _scaled_masks
            Bn, t, shapes = x.shape[0], x.shape[1], []
            for l in range(self.layers):
                t = getattr(self, 'layer' + str(l)).out_len(t)
                shapes.append((Bn, t, self.dim[l + 1]))
            shapes += [(Bn, t, (2 if self.rnn_bid else 1) * self.rnn_dim)] * self.rnn_layers
            _masks = _scaled_masks(shapes, p, x.device)
        mk = (lambda i: _masks[i]) if _masks is not None else (lambda i: None)
        for l in range(self.layers):
            x = getattr(self, 'layer' + str(l))(x, mk(l))
        B, T, _ = x.shape
        H = self.rnn_dim
        if self.training and torch.is_grad_enabled():
            for layer in range(self.rnn_layers):
                g = lambda n, rev: getattr(self.rnn, '%s_l%d%s' % (n, layer, '_reverse' if rev else ''))
                if self.rnn_bid:
                    xp_f, xp_b = AG.conv_group(x, [g('weight_ih', False), g('weight_ih', True)], [0, 0], [None, None],
                                               biases=[g('bias_ih', False), g('bias_ih', True)])
                    x = AG.bilstm(xp_f, xp_b, g('weight_hh', False), g('bias_hh', False), g('weight_hh', True), g('bias_hh', True))
                else:                                                                        # rnn_bid: False (src/asr.py:35-37)
                    x = AG.lstm(AG.conv(x, g('weight_ih', False), g('bias_ih', False)), g('weight_hh', False), g('bias_hh', False))
                last = layer == self.rnn_layers - 1
                if last and self.layer_norm:
                    x = AG.layer_norm(x, self.norm_layer)                                    # :57-58 (before self.drop)
                if p > 0:   # inter-layer dropout of nn.LSTM, and (after the last layer) the dropout in front of the projection
                    x = AG.mask_mul(x, mk(self.layers + layer))
            return AG.conv(x, self.postnet.weight, self.postnet.bias)
        Do = (2 if self.rnn_bid else 1) * H
        for layer in range(self.rnn_layers):
            out = torch.empty(B, T, Do, device=x.device, dtype=torch.float32)
            g = lambda n, rev: getattr(self.rnn, '%s_l%d%s' % (n, layer, '_reverse' if rev else ''))
            if self.rnn_bid:
                xp = [ops.gemm(x, g('weight_ih', rev), bias=g('bias_ih', rev)) for rev in (False, True)]
                ops.lstm_seq2(xp[0], xp[1], g('weight_hh', False), g('weight_hh', True), g('bias_hh', False), g('bias_hh', True), out)
            else:
                ops.lstm_seq(ops.gemm(x, g('weight_ih', False), bias=g('bias_ih', False)), g('weight_hh', False), g('bias_hh', False),
                             out, 0, False)
            x = out
            if layer == self.rnn_layers - 1 and self.layer_norm:
                ln = self.norm_layer
                x = ops.layer_norm(x.view(-1, Do), ln.weight, ln.bias, ln.eps).view(B, T, Do)
            if p > 0:   # inter-layer dropout of nn.LSTM and the dropout in front of the projection (:62)
                x = ops.act_bwd(x.view(-1, Do), None, None, mk(self.layers + layer).view(-1, Do)).view(B, T, Do)
        return ops.gemm(x, self.postnet.weight, bias=self.postnet.bias)


class ASRPostnet(nn.Module):
    """2-layer BiLSTM(latent -> latent, inter-layer dropout 0.5) -> dropout 0.5 -> Linear(2 latent -> vocab) -> log_softmax.
    ref: src/asr.py:67-80 (state_dict keys `rnn.*`, `linear.*`).  Runs on the encoder's sequence kernels; differentiable whenever
    gradients are enabled (the two dropouts only in training mode; `_masks` = explicit scaled masks for tests)."""

    def __init__(self, latent_dim, vocab_size):
        super().__init__()
        self.rnn = nn.LSTM(latent_dim, latent_dim, num_layers=2, dropout=0.5, bidirectional=True, batch_first=True)
        self.dropout = nn.Dropout(0.5)
        self.linear = nn.Linear(latent_dim * 2, vocab_size)

    def forward(self, x, _masks=None):
        x = x.contiguous()
        B, T, _ = x.shape
        H = self.rnn.hidden_size
        p = 0.5 if self.training else 0.0
        grad = torch.is_grad_enabled()
        for layer in range(2):
            g = lambda n, rev: getattr(self.rnn, '%s_l%d%s' % (n, layer, '_reverse' if rev else ''))
            if grad:
                xp_f = AG.conv(x, g('weight_ih', False), g('bias_ih', False))
                xp_b = AG.conv(x, g('weight_ih', True), g('bias_ih', True))
                x = AG.bilstm(xp_f, xp_b, g('weight_hh', False), g('bias_hh', False), g('weight_hh', True), g('bias_hh', True))
            else:
                out = torch.empty(B, T, 2 * H, device=x.device, dtype=torch.float32)
                xp = [ops.gemm(x, g('weight_ih', rev), bias=g('bias_ih', rev)) for rev in (False, True)]
                ops.lstm_seq2(xp[0], xp[1], g('weight_hh', False), g('weight_hh', True), g('bias_hh', False), g('bias_hh', True), out)
                x = out
            if p > 0 or _masks is not None:   # nn.LSTM's inter-layer dropout after layer 0, self.dropout after layer 1
                if _masks is not None:
                    m = _masks[layer]
                else:
                    from .module import _scaled_mask
                    m = _scaled_mask(tuple(x.shape), p, x.device)
                x = AG.mask_mul(x, m) if grad else ops.act_bwd(x.reshape(-1, 2 * H), None, None, m.reshape(-1, 2 * H)).view(B, T, 2 * H)
        if grad:
            return AG.log_softmax(AG.conv(x, self.linear.weight, self.linear.bias))
        return ops.log_softmax(ops.gemm(x, self.linear.weight, bias=self.linear.bias))
