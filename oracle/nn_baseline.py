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


# ------------------------------------------------------------------------------------------------------------------------------
# Whole Tacotron2.forward and the paired TTS training step from stock torch.nn modules (round 3: CPU baselines of the secondary
# bench lines -- BASELINE.md section 3, regions b-d).  Same rule as above: the ATen kernels the reference would hit.
def _copy(dst: nn.Module, W: Dict[str, Tensor], prefix: str) -> nn.Module:
    sd = dst.state_dict()
    for k in sd:
        if k.endswith('num_batches_tracked'):
            continue
        sd[k].copy_(W[prefix + k])
    return dst


class NNTacotron2(nn.Module):
    """src/tts.py:12-51 with spkr_embed_mode 'adaIN': Encoder (3 x [Conv1d k5 -> BatchNorm1d -> ReLU] -> BiLSTM, src/module.py:410-462),
    the decoder above (free running) or its teacher-forced form (src/module.py:140-214 with tf_rate = 1), CBHG(K = 8) + Linear
    (src/module.py:558-622, src/tts.py:29-34)."""

    def __init__(self, W: Dict[str, Tensor], hp: dict):
        super().__init__()
        self.hp = hp
        self.enc_convs, self.enc_bns = nn.ModuleList(), nn.ModuleList()
        i = 0
        while ('encoder.convs.%d.0.conv.weight' % i) in W:
            w = W['encoder.convs.%d.0.conv.weight' % i]
            self.enc_convs.append(_copy(nn.Conv1d(w.shape[1], w.shape[0], w.shape[2], padding=(w.shape[2] - 1) // 2), W,
                                        'encoder.convs.%d.0.conv.' % i))
            self.enc_bns.append(_copy(nn.BatchNorm1d(w.shape[0]), W, 'encoder.convs.%d.1.' % i))
            i += 1
        n_layers = 1
        while ('encoder.lstm.weight_ih_l%d' % n_layers) in W:
            n_layers += 1
        w_ih = W['encoder.lstm.weight_ih_l0']
        self.enc_lstm = _copy(nn.LSTM(w_ih.shape[1], w_ih.shape[0] // 4, n_layers, batch_first=True, bidirectional=True), W, 'encoder.lstm.')
        self.decoder = NNDecoder(W, hp)
        self.q_drop, self.d_drop = hp.get('query_dropout', 0.0), hp.get('dec_dropout', 0.0)
        p = 'postnet.0.'
        self.banks, self.bank_bns = nn.ModuleList(), nn.ModuleList()
        k = 0
        while (p + 'conv1d_banks.%d.conv1d.weight' % k) in W:
            w = W[p + 'conv1d_banks.%d.conv1d.weight' % k]
            self.banks.append(_copy(nn.Conv1d(w.shape[1], w.shape[0], w.shape[2], padding=w.shape[2] // 2, bias=False), W,
                                    p + 'conv1d_banks.%d.conv1d.' % k))
            self.bank_bns.append(_copy(nn.BatchNorm1d(w.shape[0], momentum=0.99, eps=1e-3), W, p + 'conv1d_banks.%d.bn.' % k))
            k += 1
        self.pool = nn.MaxPool1d(kernel_size=2, stride=1, padding=1)
        self.projs, self.proj_bns = nn.ModuleList(), nn.ModuleList()
        for j in range(2):
            w = W[p + 'conv1d_projs.%d.conv1d.weight' % j]
            self.projs.append(_copy(nn.Conv1d(w.shape[1], w.shape[0], 3, padding=1, bias=False), W, p + 'conv1d_projs.%d.conv1d.' % j))
            self.proj_bns.append(_copy(nn.BatchNorm1d(w.shape[0], momentum=0.99, eps=1e-3), W, p + 'conv1d_projs.%d.bn.' % j))
        self.pre_highway = _linear(W[p + 'pre_highway_proj.weight'])
        self.hw_H, self.hw_T = nn.ModuleList(), nn.ModuleList()
        j = 0
        while (p + 'highways.%d.H.weight' % j) in W:
            self.hw_H.append(_linear(W[p + 'highways.%d.H.weight' % j], W[p + 'highways.%d.H.bias' % j]))
            self.hw_T.append(_linear(W[p + 'highways.%d.T.weight' % j], W[p + 'highways.%d.T.bias' % j]))
            j += 1
        g_ih = W[p + 'gru.weight_ih_l0']
        self.gru = _copy(nn.GRU(g_ih.shape[1], g_ih.shape[0] // 3, 1, batch_first=True, bidirectional=True), W, p + 'gru.')
        self.last = _linear(W['postnet.1.weight'], W['postnet.1.bias'])

    def encode(self, txt_embed: Tensor) -> Tensor:
        x = txt_embed.transpose(1, 2)                                           # :446
        for conv, bn in zip(self.enc_convs, self.enc_bns):
            x = F.relu(bn(conv(x)))                                             # :448-452 (dropout p = 0 in every shipped config)
        y, _ = self.enc_lstm(x.transpose(1, 2))                                 # :458-460
        return y

    def postnet(self, mel: Tensor) -> Tensor:
        T = mel.shape[1]
        x = mel.transpose(1, 2)                                                 # :594
        x = torch.cat([bn(F.relu(conv(x)))[:, :, :T] for conv, bn in zip(self.banks, self.bank_bns)], dim=1)    # :597-598
        x = self.pool(x)[:, :, :T]                                              # :600
        x = self.proj_bns[0](F.relu(self.projs[0](x)))                          # :602-603 (conv -> ReLU -> BN; the last one without ReLU)
        x = self.proj_bns[1](self.projs[1](x))
        x = self.pre_highway(x.transpose(1, 2)) + mel                           # :607-609
        for H, Tg in zip(self.hw_H, self.hw_T):                                 # :551-554
            t = torch.sigmoid(Tg(x))
            x = F.relu(H(x)) * t + x * (1.0 - t)
        y, _ = self.gru(x)                                                      # :617
        return self.last(y)                                                     # src/tts.py:34

    def forward(self, txt_embed: Tensor, frames: int, spkr_embed: Tensor, seed: int = 0):
        """free-running inference: (mel, linear, align, stop)                  src/tts.py:36-51"""
        mel, align, stop = self.decoder(self.encode(txt_embed), frames, spkr_embed, seed=seed)
        return mel, self.postnet(mel), align, stop

    def forward_teacher(self, txt_embed: Tensor, teacher: Tensor, spkr_embed: Tensor):
        """tf_rate = 1 (every shipped config): the prenet over the whole teacher at once, every step fed the previous teacher
        group; hidden-state dropouts active in training mode                    src/module.py:166-206"""
        d = self.decoder
        memory = self.encode(txt_embed)
        B, L, E = memory.shape
        r, n_mels = d.r, d.n_mels
        steps = teacher.shape[1] // r
        Q, D = d.query_rnn.hidden_size, d.dec_rnn.hidden_size
        z = lambda *s: memory.new_zeros(*s)
        h_q, c_q, h_d, c_d = z(B, Q), z(B, Q), z(B, D), z(B, D)
        w, w_cum, ctx = z(B, L), z(B, L), z(B, E)
        pm = d.memory_layer(memory)
        tpre = d.prenet(teacher.reshape(B, steps, r * n_mels))                  # :178-179
        dec_in = d.prenet(z(B, r * n_mels))
        ada_s, ada_m = F.relu(d.ada_std(spkr_embed)), d.ada_mean(spkr_embed)
        mels, aligns, stops = [], [], []
        for t in range(steps):
            h_q, c_q = d.query_rnn(torch.cat([dec_in, ctx], dim=-1), (h_q, c_q))
            h_q = F.dropout(h_q, self.q_drop, self.training)                    # :230
            pq = d.query_layer(h_q).unsqueeze(1)
            loc = d.loc_linear(d.loc_conv(torch.stack([w, w_cum], dim=1)).transpose(1, 2))
            e = d.v(torch.tanh(pq + loc + pm)).squeeze(-1)
            w = F.softmax(e, dim=1)
            ctx = torch.bmm(w.unsqueeze(1), memory).squeeze(1)
            w_cum = w_cum + w
            h_d, c_d = d.dec_rnn(torch.cat([ctx, ada_s * (h_q - ada_m)], dim=-1), (h_d, c_d))
            h_d = F.dropout(h_d, self.d_drop, self.training)                    # :279
            y = torch.cat([h_d, ctx], dim=-1)
            mels.append(d.proj(y).view(B, r, n_mels))
            aligns.append(w)
            stops.append(d.gate(y).repeat(1, r))
            dec_in = tpre[:, t]                                                 # :200-203 (teacher frame group t feeds step t + 1)
        mel = torch.cat(mels, dim=1)
        return mel, self.postnet(mel), torch.stack(aligns, dim=1), torch.cat(stops, dim=1)


def train_step(model: 'NNTacotron2', opt: torch.optim.Optimizer, txt_embed: Tensor, spkr_embed: Tensor, mel: Tensor, linear: Tensor,
               freq_loss, clip: float = 5.0):
    """forward + freq_loss(mel) + freq_loss(linear) + backward + clip_grad_norm_(5.0) + optimizer step
    (bin/train_vqvae.py:219-223,270; src/solver.py:138-151), the codebook lookup and speaker table left out (negligible)"""
    opt.zero_grad()
    mel_p, lin_p, _, _ = model.forward_teacher(txt_embed, mel, spkr_embed)
    loss = freq_loss(mel_p, mel) + freq_loss(lin_p, linear)
    loss.backward()
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
    opt.step()
    return float(loss.detach()), float(gn)


# ------------------------------------------------------------------------------------------------------------------------------
# The speech <-> text cycles of VqvaeTrainer.exec from stock torch.nn modules (round 6: CPU baseline of `bench.py --workload cycle`).
class NNCtc(nn.Module):
    """CTC speech encoder (src/asr.py:5-64) with ConvLayer (src/module.py:627-648) as nn.Conv1d / nn.BatchNorm1d / nn.LSTM / nn.Linear"""

    def __init__(self, W: Dict[str, Tensor], cfg: dict, prefix: str = 'asr.'):
        super().__init__()
        self.cfg = cfg
        self.convs, self.bns = nn.ModuleList(), nn.ModuleList()
        for l, (k, s) in enumerate(zip(cfg['kernel'], cfg['stride'])):
            w = W[prefix + 'layer%d.conv.weight' % l]
            self.convs.append(_copy(nn.Conv1d(w.shape[1], w.shape[0], k, s, padding=1 if k != 1 else 0), W, prefix + 'layer%d.conv.' % l))
            self.bns.append(_copy(nn.BatchNorm1d(w.shape[0]), W, prefix + 'layer%d.bn.' % l) if cfg['batch_norm'] else None)
        self.act = getattr(torch, cfg['activation'].lower())
        self.drop = nn.Dropout(cfg['dropout'])
        w_ih = W[prefix + 'rnn.weight_ih_l0']
        self.rnn = _copy(nn.LSTM(w_ih.shape[1], cfg['rnn_dim'], cfg['rnn_layers'], dropout=cfg['dropout'], batch_first=True,
                                 bidirectional=bool(cfg['rnn_bid'])), W, prefix + 'rnn.')
        self.norm = _copy(nn.LayerNorm(W[prefix + 'postnet.weight'].shape[1]), W, prefix + 'norm_layer.') if cfg['layer_norm'] else None
        self.postnet = _linear(W[prefix + 'postnet.weight'], W[prefix + 'postnet.bias'])

    def forward(self, x: Tensor) -> Tensor:
        x = x.transpose(1, 2)                                                   # src/asr.py:49
        for conv, bn, res in zip(self.convs, self.bns, self.cfg['residual']):   # ConvLayer.forward, src/module.py:638-648
            y = conv(x)
            if bn is not None:
                y = bn(y)
            y = self.act(y)
            if res:
                y = y + x
            x = self.drop(y)
        y, _ = self.rnn(x.transpose(1, 2))                                      # :56
        if self.norm is not None:
            y = self.norm(y)
        return self.postnet(self.drop(y))                                       # :62-63


class NNVqvae(nn.Module):
    """VQVAE with the L2 codebook (src/vqvae.py, src/embed.py:57-147; stop_grad = True, skip_prob = 0, no attribute table: the
    synthetic-weight configuration bench.py runs)"""

    def __init__(self, W: Dict[str, Tensor], model_cfg: dict, n_mels: int):
        super().__init__()
        hp = dict(model_cfg['decoder']['decoder'], n_mels=n_mels)
        self.asr = NNCtc(W, model_cfg['encoder'])
        self.table = nn.Parameter(W['codebook.learnable_table'].clone())
        self.register_buffer('temp', W['codebook.temp'].clone())
        self.spkr = nn.Embedding.from_pretrained(W['spkr_embed.weight'].clone(), freeze=False)
        self.tts = NNTacotron2({k[4:]: v for k, v in W.items() if k.startswith('tts.')}, hp)
        self.max_frames_per_phn = model_cfg['max_frames_per_phn']

    def codebook(self, x: Tensor):
        """L2Embedding.forward, src/embed.py:105-147"""
        flat = x.reshape(-1, x.shape[-1])
        sim = -((flat.pow(2).sum(-1, keepdim=True) + self.table.pow(2).sum(-1)) - 2 * flat.matmul(self.table.t()))
        p = (torch.relu(self.temp) * sim).view(x.shape[0], x.shape[1], -1).softmax(dim=-1)
        code = F.embedding(p.argmax(dim=-1), self.table)
        return p, x + code - x.detach()

    def mean_forward(self, p_code: Tensor, latent: Tensor):
        """src/vqvae.py:218-257 as the reference runs it: one host list per utterance"""
        T = latent.shape[1]
        out, lens = [], []
        for b, idx_seq in enumerate(p_code.argmax(dim=-1)):
            idx_seq = idx_seq.tolist()
            last_idx, last_pos, cur = idx_seq[0], 0, []
            for t, idx in enumerate(idx_seq):
                if last_idx != idx or (t - last_pos) > self.max_frames_per_phn:
                    if last_idx != 0:
                        cur.append(latent[b, last_pos:t].mean(dim=0))
                    last_idx, last_pos = idx, t
            if last_idx != 0:
                cur.append(latent[b, last_pos:].mean(dim=0) if last_pos != T - 1 else latent[b, t])
            if not cur:
                return None
            lens.append(len(cur))
            out.append(torch.stack(cur, dim=0))
        return nn.utils.rnn.pad_sequence(out, batch_first=True), torch.LongTensor(lens)


def _pad_cat(a: Tensor, b: Tensor) -> Tensor:
    T = max(a.shape[1], b.shape[1])
    return torch.cat([F.pad(a, (0, 0, 0, T - a.shape[1])), F.pad(b, (0, 0, 0, T - b.shape[1]))], dim=0)


def cycle_step(model: 'NNVqvae', opt: torch.optim.Optimizer, kind: str, pair, unpair, freq_loss, hparas: dict, clip: float = 5.0):
    """one iteration of VqvaeTrainer.exec (bin/train_vqvae.py:124-270) with unpair_text_weight = 0 (the shipped configurations):
    `speech_first` = speech_to_text(paired || unpaired aug_mel) -> mean_forward -> text_to_speech(paired text || merged latents),
    `text_first` = text_to_speech(paired) -> speech_to_text(paired); CTC + freq losses, backward, clip, optimiser step"""
    mel, aug, linear, text, sid = pair
    opt.zero_grad()
    Bp = mel.shape[0]
    lat_p = F.embedding(text, model.table)
    use_un = kind == 'speech_first' and unpair is not None
    if kind == 'speech_first':
        x = _pad_cat(aug, unpair[1]) if use_un else aug
        p, q = model.codebook(model.asr(x))
        merged = model.mean_forward(p[Bp:], q[Bp:]) if use_un else None
        use_un = merged is not None
    if use_un:
        lat = _pad_cat(lat_p, merged[0])
        teacher = _pad_cat(mel, unpair[0])
        spk = torch.cat([model.spkr(sid), model.spkr(unpair[4])], dim=0)
    else:
        lat, teacher, spk = lat_p, mel, model.spkr(sid)
    mel_p, _, _, _ = model.tts.forward_teacher(lat, teacher, spk)
    lin_p = model.tts.postnet(mel_p.detach())                                   # separate_postnet: true (src/tts.py:47-50)
    if kind != 'speech_first':
        p, _ = model.codebook(model.asr(aug))
    pp = p[:Bp]
    tgt_len = (text != 0).sum(dim=-1)
    ctc = F.ctc_loss((pp + 1e-10).transpose(0, 1).log(), text[text != 0], torch.full((Bp,), pp.shape[1], dtype=torch.long), tgt_len)
    Tp = mel.shape[1]
    loss = float(hparas.get('asr_weight', 1.0)) * ctc + float(hparas.get('tts_weight', 1.0)) * (
        freq_loss(mel_p[:Bp, :Tp], mel) + freq_loss(lin_p[:Bp, :Tp], linear))
    if use_un:
        Tu = unpair[0].shape[1]
        loss = loss + float(hparas.get('unpair_speech_weight', 10.0)) * (freq_loss(mel_p[Bp:, :Tu], unpair[0]) + freq_loss(lin_p[Bp:, :Tu], unpair[2]))
    loss.backward()
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
    opt.step()
    return float(loss.detach()), float(gn)
