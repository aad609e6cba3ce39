"""The two training cycles of VqvaeTrainer.exec (bin/train_vqvae.py:124-270) at CONFIG dimensions -- config/semi-single-spkr-paired-data.yaml:
speech encoder 6 x 512 channels with the stride-2 layer, 2-layer BiLSTM(256), L2 codebook, run-length merge, the full TTS branch, B = 8 + 8,
256 -> 258 frames -- through `VqvaeTrainer.speech_first_step / text_first_step` (the C ABI underneath) against fp32 CPU autograd through the
oracle replaying the same dropout masks.  The tiny-dimension goldens recorded from the real reference pin the oracle's composition
(tests/test_gpu_grad.py::test_speech_first_step_against_reference_golden, ::test_text_first_step_against_reference_golden); these tests
carry that to the sizes the kernels are tuned for.  VQ indices must agree bit for bit (they fix the merged text length)."""
import os
import sys
from argparse import Namespace

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import REPO, coin_source, masks_to, maxdiff, report, split_masks   # noqa: E402

pytestmark = pytest.mark.gpu

from oracle import asr_oracle as AO   # noqa: E402
from oracle import tts_oracle as O    # noqa: E402
from oracle import vq_oracle as VQ    # noqa: E402

EPS = 1e-10


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs an MI355X')
    from semi_tts_amd import _lib
    _lib.load()
    return torch.device('cuda', 0)


def _config():
    import yaml
    cfg = yaml.safe_load(open(os.path.join(REPO, 'config', 'semi-single-spkr-paired-data.yaml')))
    cfg['model']['codebook'].update(phn_attr_pth='', proj_attr=None)          # the attribute csv lives in the reference tree
    return cfg


def _pad_cat(a, b):
    T = max(a.shape[1], b.shape[1])
    return torch.cat([F.pad(a, (0, 0, 0, T - a.shape[1])), F.pad(b, (0, 0, 0, T - b.shape[1]))], dim=0)


def _ctc(p, text):
    """compute_ctcloss with actual_len = False (bin/train_vqvae.py:430-444): torch.nn.CTCLoss() on log(p + EPS), every frame counts"""
    Bn, S = p.shape[0], p.shape[1]
    return F.ctc_loss((p + EPS).transpose(0, 1).log(), text[text != 0], torch.full((Bn,), S, dtype=torch.long), (text != 0).sum(dim=-1))


class _Oracle:
    """the weights of a VQVAE as autograd leaves + the pieces of the two cycles (every call cites what it follows)"""

    def __init__(self, model, cfg):
        W = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        leaf = lambda k, v: v.requires_grad_(v.is_floating_point() and 'running_' not in k)
        self.Wt = {k[4:]: leaf(k, v) for k, v in W.items() if k.startswith('tts.')}
        self.Wa = {k[4:]: leaf(k, v) for k, v in W.items() if k.startswith('asr.')}
        self.table = W['codebook.learnable_table'].requires_grad_()
        self.Wc = {'learnable_table': self.table, 'temp': W['codebook.temp']}
        self.spk = W['spkr_embed.weight'].requires_grad_()
        m = cfg['model']
        self.enc_cfg = m['encoder']
        self.hp = dict(m['decoder']['decoder'], n_mels=80, enc_dropout=m['decoder']['encoder']['enc_dropout'])
        self.max_frames = m['max_frames_per_phn']
        self.sr, self.n_mels = cfg['data']['audio']['sample_rate'], cfg['data']['audio']['num_mels']
        self.drop = O.DropoutSource('rng', generator=torch.Generator().manual_seed(23))

    def speech_to_text(self, aug, n_real=0):
        """VQVAE.speech_to_text up to the codebook (src/vqvae.py:106-119); returns (p_code, idx, new_latent, the masks drawn)"""
        n0 = len(self.drop.used)
        enc = AO.ctc_forward(self.Wa, aug, self.enc_cfg, training=True, drop=self.drop)
        p, idx, lat, _ = VQ.l2_forward(self.Wc, enc, n_real)
        return p, idx, lat, self.drop.used[n0:]

    def tts(self, lat, teacher, spk, umax=None):
        """Tacotron2.forward with separate_postnet (src/tts.py:36-51); returns (mel, linear, the masks drawn)"""
        n0 = len(self.drop.used)
        mem = O.encoder_forward(self.Wt, lat, 'encoder.', True, self.hp['enc_dropout'], self.drop, None)
        mel, al, _ = O.decoder_forward(self.Wt, mem, teacher, spk, self.hp, 1.0, umax, True, self.drop, lambda: 0.0)
        lin = O.postnet_forward(self.Wt, mel.detach(), True, None)
        return mel, lin, self.drop.used[n0:]

    def freq(self, p, l):
        return O.freq_loss(p, l, self.sr, self.n_mels)

    def grads(self):
        g = {'tts.' + k: v.grad for k, v in self.Wt.items() if v.requires_grad and v.grad is not None}
        g.update({'asr.' + k: v.grad for k, v in self.Wa.items() if v.requires_grad and v.grad is not None})
        g['codebook.learnable_table'] = self.table.grad
        g['spkr_embed.weight'] = self.spk.grad
        return {k: v for k, v in g.items() if v is not None}


def _trainer(cfg, model, hparas=None):
    from semi_tts_amd.optim import Optimizer
    from semi_tts_amd.solver import VqvaeTrainer
    h = dict(cfg['hparas'], **(hparas or {}))
    tr = VqvaeTrainer(dict(cfg, hparas=h), Namespace(vocab_size=43, n_spkr=109, verbose=False, max_step=1), 'train')
    tr.model = model
    tr.optimizer = Optimizer(model.parameters(), h['optimizer'], h['lr'], h['lr_scheduler'], tf_start=h['tf_start'], tf_end=h['tf_end'],
                             tf_step=h['tf_step'])
    grads = {}
    orig = tr.clip_grad_norm_

    def spy(params, max_norm):
        for k, p in model.named_parameters():
            if p.grad is not None:
                grads[k] = p.grad.detach().clone()
        return orig(list(params), max_norm)
    tr.clip_grad_norm_ = spy
    return tr, grads


def _compare(name, ref_g, grads, min_n):
    """three bounds per gradient tensor (see test_c2_training_step_against_oracle_autograd: a ReLU / max-pool / tanh-saturated element on
    the other side of a kink changes ONE term of a sum): 99.9th-percentile element error and relative L2 tightly, the worst element loosely"""
    worst, worst_k, worst_p999, worst_l2, n = 0.0, '', 0.0, 0.0, 0
    gmax = max(float(g.abs().max()) for g in ref_g.values())
    for k, g in ref_g.items():
        assert k in grads, 'missing gradient for ' + k
        scale = float(g.abs().max())
        if scale < 1e-6 * gmax:          # analytically zero (a conv bias in front of a batch-statistics BatchNorm): round-off on both sides
            assert float(grads[k].abs().max()) < 1e-4 * gmax, k
            continue
        d = (grads[k].detach().cpu().double() - g.double()).abs().flatten()
        e = float(d.max()) / scale
        p999 = float(d.kthvalue(max(1, int(0.999 * d.numel())))[0]) / scale
        l2 = float(d.norm() / g.double().norm())
        worst_p999, worst_l2 = max(worst_p999, p999), max(worst_l2, l2)
        n += 1
        if e > worst:
            worst, worst_k = e, k
        assert p999 < 5e-3 and l2 < 5e-3 and e < 0.1, (k, e, p999, l2)
    report(name + '_grads', worst=worst, worst_k=worst_k, worst_p999=worst_p999, worst_rel_l2=worst_l2, n=n)
    assert n >= min_n, n


def _model(cfg, dev, seed=1234):
    from semi_tts_amd.synthetic import load_synthetic
    from semi_tts_amd.vqvae import VQVAE
    m = VQVAE(80, 1025, 43, 109, **cfg['model'])
    load_synthetic(m, seed)
    return m.to(dev).train()


def _batches(Bp, Bu, frames=256):
    from semi_tts_amd.synthetic import synthetic_cycle_batch
    return synthetic_cycle_batch(Bp, frames, 3, seed=17), synthetic_cycle_batch(Bu, frames, 3, seed=29)


def test_speech_first_cycle_at_config_size_against_oracle_autograd(dev):
    """speech -> text -> speech with the unpaired batch (bin/train_vqvae.py:159-176,208-233): CTC encoder on paired || unpaired aug_mel,
    VQ search, run-length merge of the unpaired part, the TTS branch on paired text || merged latents (L = the longest merged sequence),
    CTC + freq_loss(paired) + 10 freq_loss(unpaired); every dropout (speech encoder 0.5, prenet 0.5, cells 0.1) and every batch-statistics
    BatchNorm active."""
    cfg = _config()
    Bp = Bu = 8
    model = _model(cfg, dev)
    pair, unpair = _batches(Bp, Bu)
    mel, aug, linear, text, sid = pair
    umel, uaug, ulin, _, usid = unpair
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    orc = _Oracle(model, cfg)
    p, idx, lat, asr_masks = orc.speech_to_text(_pad_cat(aug, uaug))
    merged = VQ.mean_forward_torch(idx[Bp:], lat[Bp:], orc.max_frames)
    assert merged is not None
    all_lat = _pad_cat(VQ.l2_inference(orc.Wc, text), merged[0])
    mel_r, lin_r, tts_masks = orc.tts(all_lat, _pad_cat(mel, umel), orc.spk[torch.cat([sid, usid])])
    Tp, Tu = mel.shape[1], umel.shape[1]
    asr_loss = _ctc(p[:Bp], text)
    tts_loss = orc.freq(mel_r[:Bp, :Tp], mel) + orc.freq(lin_r[:Bp, :Tp], linear)
    un_loss = orc.freq(mel_r[Bp:, :Tu], umel) + orc.freq(lin_r[Bp:, :Tu], ulin)
    h = cfg['hparas']
    total = h['asr_weight'] * asr_loss + h['tts_weight'] * tts_loss + h['unpair_speech_weight'] * un_loss
    total.backward()
    ref_g = orc.grads()
    gn_ref = float(torch.sqrt(sum((g.double() ** 2).sum() for g in ref_g.values())))

    tr, grads = _trainer(cfg, model)
    tr.step = 2                                   # an even step past unpair_speech_start_step (0): the unpaired term counts (:232)
    B, steps = Bp + Bu, max(Tp, Tu) // 3
    masks = masks_to(split_masks(tts_masks, orc.hp, True, 1.0, B, B, steps, list(range(steps)), orc.hp['prenet_dim']), dev)
    d = lambda t: t.to(dev)
    st = tr.speech_first_step(d(mel), d(aug), d(linear), d(text), d(sid), unpair_mel=d(umel), unpair_aug_mel=d(uaug), unpair_linear=d(ulin),
                              unpair_sid=d(usid), _masks=masks, _asr_masks=[d(m) for m in asr_masks])
    assert torch.equal(model.codebook.last_idx.cpu(), idx)                      # VQ indices: bit-exact (2 048 vectors, 43 codes)
    assert st['unpair_text_len'] == merged[0].shape[1]
    errs = dict(asr_loss=abs(st['asr_loss'] - float(asr_loss)), tts_loss=abs(st['tts_loss'] - float(tts_loss)),
                unpair_speech_loss=abs(st['unpair_speech_loss'] - float(un_loss)), loss=abs(st['loss'] - float(total)),
                loss_ref=float(total), grad_norm=st['grad_norm'], grad_norm_ref=gn_ref, merged_len=int(merged[0].shape[1]))
    report('speech_first_config_size', **errs)
    for k in ('asr_loss', 'tts_loss', 'unpair_speech_loss', 'loss'):
        assert errs[k] < 2e-5 * max(1.0, abs(errs['loss_ref'])), (k, errs)
    assert abs(st['grad_norm'] - gn_ref) < 1e-3 * gn_ref
    _compare('speech_first_config_size', ref_g, grads, 120)


@pytest.mark.parametrize('with_unpaired_text', [False, True])
def test_text_first_cycle_at_config_size_against_oracle_autograd(dev, with_unpaired_text):
    """text -> speech -> text (bin/train_vqvae.py:186-205,208-224,234-250).  Without unpaired text (unpair_text_weight 0: every shipped
    configuration) the paired batch goes through the TTS branch and the speech encoder; with it (weight 1) the unpaired rows decode from
    their own outputs, the detached prediction is quantised next to the paired aug_mel with the table detached for the fake part."""
    cfg = _config()
    Bp = 8
    Bu = 8 if with_unpaired_text else 0
    model = _model(cfg, dev, seed=77)          # (seeds whose speech encoder spreads over the codes: 321 / 322 collapse onto the blank)
    pair, unpair = _batches(Bp, 8)
    mel, aug, linear, text, sid = pair
    _, _, _, utext, usid = unpair
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    orc = _Oracle(model, cfg)
    r, Tp = 3, mel.shape[1]
    lat_p = VQ.l2_inference(orc.Wc, text)
    if with_unpaired_text:
        umax = int(6.0 * utext.shape[1])
        umax += umax % r                                                        # VQVAE.text_to_speech :158-160
        all_lat = _pad_cat(lat_p, VQ.l2_inference(orc.Wc, utext))
        mel_r, lin_r, tts_masks = orc.tts(all_lat, mel, orc.spk[torch.cat([sid, usid])], umax)
        upm = mel_r[Bp:, :umax].detach()                                        # :201-202
        p, idx, _, asr_masks = orc.speech_to_text(_pad_cat(aug, upm), n_real=Bp)
    else:
        umax = None
        mel_r, lin_r, tts_masks = orc.tts(lat_p, mel, orc.spk[sid])
        p, idx, _, asr_masks = orc.speech_to_text(aug)
    asr_loss = _ctc(p[:Bp], text)
    tts_loss = orc.freq(mel_r[:Bp, :Tp], mel) + orc.freq(lin_r[:Bp, :Tp], linear)
    total = asr_loss + tts_loss
    if with_unpaired_text:
        ut_loss = _ctc(p[Bp:], utext)
        total = total + 1.0 * ut_loss
    total.backward()
    ref_g = orc.grads()
    gn_ref = float(torch.sqrt(sum((g.double() ** 2).sum() for g in ref_g.values())))

    tr, grads = _trainer(cfg, model, dict(unpair_text_weight=1.0) if with_unpaired_text else None)
    tr.step = 3
    B = Bp + Bu
    from semi_tts_amd.module import plan_decode
    steps, src = plan_decode(False, Tp, Bp, B, r, 1.0, orc.hp['drop_dec_in'], umax, coin_source([0.0] * 1000))
    masks = masks_to(split_masks(tts_masks, orc.hp, True, 1.0, B, Bp, steps, src, orc.hp['prenet_dim']), dev)
    d = lambda t: t.to(dev)
    kw = dict(unpair_text=d(utext), unpair_sid=d(usid)) if with_unpaired_text else {}
    st = tr.text_first_step(d(mel), d(aug), d(linear), d(text), d(sid), _masks=masks, _asr_masks=[d(m) for m in asr_masks], **kw)
    assert torch.equal(model.codebook.last_idx.cpu(), idx)                      # VQ indices: bit-exact
    name = 'text_first_config_size' + ('_unpaired_text' if with_unpaired_text else '')
    errs = dict(asr_loss=abs(st['asr_loss'] - float(asr_loss)), tts_loss=abs(st['tts_loss'] - float(tts_loss)),
                loss=abs(st['loss'] - float(total)), loss_ref=float(total), grad_norm=st['grad_norm'], grad_norm_ref=gn_ref)
    if with_unpaired_text:
        errs['unpair_text_loss'] = abs(st['unpair_text_loss'] - float(ut_loss))
    report(name, **errs)
    for k in errs:
        if k.endswith('loss'):
            assert errs[k] < 2e-5 * max(1.0, abs(errs['loss_ref'])), (k, errs)
    assert abs(st['grad_norm'] - gn_ref) < 1e-3 * gn_ref
    _compare(name, ref_g, grads, 120)


def test_cycle_steps_asynchronous_statistics_equal_the_synchronous_ones(dev):
    """VqvaeTrainer with async_stats (no host read of a loss or the gradient norm inside a step; the merged lengths of mean_forward are the
    one read of a speech-first step) takes the same four alternating steps as the synchronous trainer: same statistics, same weights bit for
    bit -- on a config-size model at B = 4 + 4, 66 frames."""
    cfg = _config()
    from semi_tts_amd.synthetic import synthetic_cycle_batch
    pair = [t.to(dev) for t in synthetic_cycle_batch(4, 64, 3, seed=5)]
    unpair = [t.to(dev) for t in synthetic_cycle_batch(4, 64, 3, seed=6)]
    outs = []
    for asyn in (False, True):
        model = _model(cfg, dev, seed=77)
        tr, _ = _trainer(cfg, model)
        tr.clip_grad_norm_ = type(tr).clip_grad_norm_          # (no spy: the asynchronous path hands the device norm to the guarded Adam)
        tr.async_stats = asyn
        torch.manual_seed(11)
        tr.step = 2
        log = [tr.cycle_step(pair, unpair if tr.cycle_kind(tr.step)[1] else None) for _ in range(4)]
        tr.drain_stats()
        torch.cuda.synchronize()
        outs.append(([{k: float(st[k]) for k in ('loss', 'asr_loss', 'tts_loss', 'grad_norm')} for st in log],
                     [st['kind'] for st in log], {k: v.detach().clone() for k, v in model.state_dict().items()}))
    (s0, k0, w0), (s1, k1, w1) = outs
    assert k0 == k1 == ['speech_first', 'text_first', 'speech_first', 'text_first']
    assert s0 == s1, (s0, s1)
    for k in w0:
        assert torch.equal(w0[k], w1[k]), k


def test_cycle_steps_with_the_postnet_branch_on_a_second_stream_equal_the_one_stream_steps(dev):
    """separate_postnet: the linear-spectrogram losses and the CBHG backward of both cycle kinds on the second stream (Tacotron2.postnet_side,
    VqvaeTrainer._side_branch) beside the main backward -- four alternating steps at B = 4 + 4: same statistics, same weights bit for bit as
    the steps on one stream."""
    cfg = _config()
    assert cfg['model']['decoder'].get('separate_postnet', False)
    from semi_tts_amd.synthetic import synthetic_cycle_batch
    pair = [t.to(dev) for t in synthetic_cycle_batch(4, 64, 3, seed=5)]
    unpair = [t.to(dev) for t in synthetic_cycle_batch(4, 64, 3, seed=6)]
    outs = []
    for side in (False, True):
        model = _model(cfg, dev, seed=77)
        tr, _ = _trainer(cfg, model)
        tr.postnet_side = side
        tr.clip_grad_norm_ = type(tr).clip_grad_norm_
        tr.async_stats = True
        torch.manual_seed(11)
        tr.step = 2
        log = [tr.cycle_step(pair, unpair if tr.cycle_kind(tr.step)[1] else None) for _ in range(4)]
        assert (model.tts.postnet_stream is not None) == side
        tr.drain_stats()
        torch.cuda.synchronize()
        outs.append(([{k: float(st[k]) for k in ('loss', 'asr_loss', 'tts_loss', 'grad_norm')} for st in log],
                     [float(st.get('unpair_speech_loss', 0.0)) for st in log], {k: v.detach().clone() for k, v in model.state_dict().items()}))
    (s0, u0, w0), (s1, u1, w1) = outs
    assert s0 == s1, (s0, s1)
    assert u0 == u1 and any(u0), (u0, u1)
    for k in w0:
        assert torch.equal(w0[k], w1[k]), k
