#!/usr/bin/env python3
"""Generate golden fixtures under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference, which never travels to the GPU
box).  It imports the reference's Python modules (with empty stubs for the two absent
third-party imports `editdistance` and `soundfile`, which nothing on the hot path calls),
instantiates them with tiny dimensions, and records inputs, weights, dropout masks,
host-RNG coin flips and outputs as .npz data.  No reference source is copied anywhere.

    python tools/gen_golden.py            # (re)writes tests/golden/*.npz
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden')

import numpy as np
import torch
import torch.nn.functional as F

for _m in ('editdistance', 'soundfile'):
    sys.modules.setdefault(_m, types.ModuleType(_m))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

from src.tts import Tacotron2 as RefTacotron2            # noqa: E402
from src.embed import L2Embedding as RefL2, SeperateEmbedding as RefSep   # noqa: E402
from src.module import Postnet as RefPostnet             # noqa: E402
from src.vqvae import VQVAE as RefVQVAE                  # noqa: E402
from src.util import freq_loss as ref_freq_loss          # noqa: E402

torch.set_num_threads(1)


class Recorder:
    """Records dropout masks (scaled) and np.random.rand draws made by the reference."""

    def __init__(self):
        self.masks, self.coins = [], []

    def __enter__(self):
        self._drop, self._rand = F.dropout, np.random.rand

        def dropout(input, p=0.5, training=True, inplace=False):
            if (not training) or p == 0.0:
                return input
            m = torch.bernoulli(torch.full_like(input, 1.0 - p)) / (1.0 - p)
            self.masks.append(m.detach().clone())
            return input * m

        def rand(*a):
            v = self._rand(*a)
            if not a:
                self.coins.append(float(v))
            return v

        F.dropout = dropout
        torch.nn.functional.dropout = dropout
        np.random.rand = rand
        return self

    def __exit__(self, *a):
        F.dropout = self._drop
        torch.nn.functional.dropout = self._drop
        np.random.rand = self._rand


def randomize_buffers(mod, gen):
    """make BN running stats / biases non-trivial so eval-mode parity means something"""
    for name, buf in mod.named_buffers():
        if name.endswith('running_mean'):
            buf.copy_(torch.randn(buf.shape, generator=gen) * 0.2)
        elif name.endswith('running_var'):
            buf.copy_(torch.rand(buf.shape, generator=gen) + 0.5)
    for name, p in mod.named_parameters():
        if name.endswith('bias') and p.dim() == 1:
            p.data.copy_(torch.randn(p.shape, generator=gen) * 0.1 + p.data)
        if '.bn.weight' in name or '.norm.weight' in name or 'norm_layer.weight' in name or (name.endswith('.1.weight') and p.dim() == 1):
            p.data.copy_(torch.rand(p.shape, generator=gen) + 0.5)


def save(name, weights, arrays, meta):
    d = {}
    for k, v in weights.items():
        d['w/' + k] = v.detach().cpu().numpy()
    for k, v in arrays.items():
        if isinstance(v, (list, tuple)):
            for i, m in enumerate(v):
                d['%s/%03d' % (k, i)] = m.detach().cpu().numpy() if torch.is_tensor(m) else np.asarray(m)
        else:
            d[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    d['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **d)
    print('%-28s %8.1f KB' % (name, os.path.getsize(path) / 1024))


TINY = dict(
    n_mels=8, linear_dim=20, in_embed_dim=12, spkr_embed_dim=8,
    paras={
        'encoder': dict(enc_n_conv=3, enc_kernel_size=5, enc_rnn_layer=1, enc_embed_dim=32, enc_dropout=0.0),
        'decoder': dict(n_frames_per_step=3, prenet_dim=16, prenet_dropout=0.5, query_rnn_dim=48, dec_rnn_dim=40,
                        query_dropout=0.1, dec_dropout=0.1, attn_dim=16, n_location_filters=4,
                        location_kernel_size=5, loc_aware=True, use_summed_weights=True, drop_dec_in=0.0),
    })


def tiny_model(seed, **over):
    torch.manual_seed(seed)
    cfg = json.loads(json.dumps(TINY))
    if '_enc_rnn_layer' in over:
        cfg['paras']['encoder']['enc_rnn_layer'] = over.pop('_enc_rnn_layer')
    if '_enc_dropout' in over:
        cfg['paras']['encoder']['enc_dropout'] = over.pop('_enc_dropout')
    cfg['paras']['decoder'].update(over)
    m = RefTacotron2(cfg['n_mels'], cfg['linear_dim'], cfg['in_embed_dim'], cfg['spkr_embed_dim'], cfg['paras'])
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        randomize_buffers(m, g)
    hp = dict(cfg['paras']['decoder'])
    hp['n_mels'] = cfg['n_mels']
    hp['enc_dropout'] = cfg['paras']['encoder']['enc_dropout']
    return m, cfg, hp


def tts_case(name, seed, B, L, teacher, tf_rate, training, teacher_bs=None, unpair_max_frame=None, **over):
    m, cfg, hp = tiny_model(seed, **over)
    m.train(training)
    g = torch.Generator().manual_seed(seed + 1)
    txt = torch.randn(B, L, cfg['in_embed_dim'], generator=g)
    spk = torch.randn(B, cfg['spkr_embed_dim'], generator=g)
    if isinstance(teacher, int):
        tch = teacher
    else:
        tch = torch.rand(teacher_bs or B, teacher[0], cfg['n_mels'], generator=g)
    w0 = {k: v.clone() for k, v in m.state_dict().items()}
    np.random.seed(seed)
    torch.manual_seed(seed + 2)
    with Recorder() as rec, torch.no_grad():
        mel, lin, align, stop = m(txt, None, tch, spk, tf_rate=tf_rate, unpair_max_frame=unpair_max_frame)
    arrays = dict(txt_embed=txt, spkr_embed=spk, mel=mel, linear=lin, align=align, stop=stop,
                  mask=rec.masks, coins=np.asarray(rec.coins, np.float64))
    if not isinstance(tch, int):
        arrays['teacher'] = tch
    if training:   # running stats after the training-mode forward
        arrays['post'] = [v for k, v in m.state_dict().items() if 'running_' in k]
        arrays['post_keys'] = np.frombuffer(json.dumps([k for k in m.state_dict() if 'running_' in k]).encode(), np.uint8)
    meta = dict(hp=hp, cfg=cfg, tf_rate=tf_rate, training=training,
                teacher=tch if isinstance(tch, int) else None, unpair_max_frame=unpair_max_frame)
    save(name, w0, arrays, meta)


def vq_cases():
    os.chdir(REF)   # phn_attr_pth is relative in the configs
    base = dict(softmax='normal', latent_dim=64, commit_weight=0, vq_weight=0, temp=1, skip_prob=0, stop_grad=True)
    g = torch.Generator().manual_seed(7)
    # native 43 x 64 table with projected phoneme attributes
    torch.manual_seed(11)
    cb = RefL2(43, False, phn_attr_pth='data/phn_attr.csv', proj_attr=16, **base).eval()
    x = torch.randn(3, 17, 64, generator=g)
    # a few exact-tie / near-tie rows: midpoints of two codes, and exact code vectors
    tab = cb.embedding.weight.detach()
    x[0, 0] = 0.5 * (tab[5] + tab[9])
    x[0, 1] = tab[12]
    x[0, 2] = 0.5 * (tab[3] + tab[4]) + 1e-7
    txt = torch.randint(0, 43, (3, 11), generator=g)
    with torch.no_grad():
        p, out, _, _ = cb(x)
        inf = cb.inference(txt)
    save('vq_l2_native', dict(cb.state_dict()), dict(x=x, p_code=p, idx=p.argmax(-1), new_latent=out,
                                                     txt=txt, inference=inf, table=tab), dict(V=43, D=64))
    # synthetic 512 x 64 table, no attributes, duplicated rows -> first-index-wins ties
    torch.manual_seed(12)
    cb = RefL2(512, False, phn_attr_pth='', proj_attr=None, **base).eval()
    with torch.no_grad():
        cb.learnable_table[300] = cb.learnable_table[20]
        cb.learnable_table[301] = cb.learnable_table[20]
    x = torch.randn(2, 33, 64, generator=g)
    x[1, 0] = cb.learnable_table[20].detach()
    x[1, 1] = cb.learnable_table[20].detach() * 0.9
    with torch.no_grad():
        p, out, _, _ = cb(x)
    save('vq_l2_512', dict(cb.state_dict()), dict(x=x, p_code=p, idx=p.argmax(-1), new_latent=out),
         dict(V=512, D=64))
    # temp = 0.25 (sharper/softer distribution changes the softmax, not the index)
    torch.manual_seed(13)
    b2 = dict(base)
    b2['temp'] = 0.25
    cb = RefL2(43, False, phn_attr_pth='data/phn_attr.csv', proj_attr=16, **b2).eval()
    x = torch.randn(2, 9, 64, generator=g) * 2
    with torch.no_grad():
        p, out, _, _ = cb(x)
    save('vq_l2_temp', dict(cb.state_dict()), dict(x=x, p_code=p, idx=p.argmax(-1), new_latent=out), dict(V=43, D=64))
    # 'seperate' bone (config/supervised.yaml)
    torch.manual_seed(14)
    cb = RefSep(43, False, phn_attr_pth='data/phn_attr.csv', proj_attr=16, **base).eval()
    x = torch.randn(3, 13, 64, generator=g)
    txt = torch.randint(0, 43, (3, 11), generator=g)
    with torch.no_grad():
        p, out, _, _ = cb(x)
        inf = cb.inference(txt)
    save('vq_seperate', dict(cb.state_dict()), dict(x=x, p_code=p, idx=p.argmax(-1), new_latent=out,
                                                    txt=txt, inference=inf), dict(V=43, D=64))
    os.chdir(REPO)


def mean_forward_case():
    import yaml
    os.chdir(REF)
    cfg = yaml.safe_load(open('config/semi-single-spkr-paired-data.yaml'))['model']
    torch.manual_seed(3)
    m = RefVQVAE(80, 1025, 43, 109, **cfg)
    os.chdir(REPO)
    g = torch.Generator().manual_seed(21)
    cases = {}
    for ci, (B, T) in enumerate([(3, 23), (2, 9), (1, 1), (2, 14)]):
        idx = torch.randint(0, 4, (B, T), generator=g)            # few symbols -> long runs, blanks (0)
        if ci == 3:
            idx[1] = 0                                            # an all-blank utterance -> None
        p = F.one_hot(idx, 43).float()
        lat = torch.randn(B, T, 64, generator=g)
        out = m.mean_forward(p, lat)
        cases['idx%d' % ci] = idx
        cases['lat%d' % ci] = lat
        if out is None:
            cases['none%d' % ci] = np.array([1])
        else:
            cases['out%d' % ci] = out[0]
            cases['len%d' % ci] = out[1]
    save('vq_mean_forward', {}, cases, dict(max_frames_per_phn=cfg['max_frames_per_phn'], n_cases=4))


def postnet_class_case():
    torch.manual_seed(31)
    m = RefPostnet(8, 16, 5, 5, 0.0).eval()
    g = torch.Generator().manual_seed(32)
    with torch.no_grad():
        randomize_buffers(m, g)
        x = torch.randn(2, 11, 8, generator=g)
        y = m(x)
    save('conv_postnet_tiny', dict(m.state_dict()), dict(x=x, y=y), {})


def postnet_class_train_case():
    """the Postnet class in TRAINING mode (batch-statistics BatchNorm, nn.Dropout(0.5) after every block, src/module.py:73): masks
    recorded in the reference's (B, C, T) layout, output, running statistics after the step, and the gradients of y.pow(2).sum()"""
    torch.manual_seed(35)
    m = RefPostnet(8, 16, 5, 5, 0.5).train()
    g = torch.Generator().manual_seed(36)
    with torch.no_grad():
        randomize_buffers(m, g)
    w0 = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.randn(3, 13, 8, generator=g).requires_grad_(True)
    with Recorder() as rec:
        y = m(x)
    y.pow(2).sum().backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    arrays = dict(x=x.detach(), y=y.detach(), dx=x.grad.clone(), masks=[t.transpose(1, 2).contiguous() for t in rec.masks])
    for k, v in grads.items():
        arrays['grad/' + k] = v
    for k, v in m.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            arrays['after/' + k] = v.clone()
    save('conv_postnet_train', w0, arrays, {})


def loss_case():
    g = torch.Generator().manual_seed(41)
    pm, lm = torch.rand(2, 12, 80, generator=g), torch.rand(2, 12, 80, generator=g)
    pl, ll = torch.rand(2, 12, 1025, generator=g), torch.rand(2, 12, 1025, generator=g)
    a = ref_freq_loss(pm, lm, 22050, 80, 'mse', True, True)
    b = ref_freq_loss(pl, ll, 22050, 80, 'mse', True, True)
    c = ref_freq_loss(pl, ll, 22050, 80, 'l1', True, True)
    save('freq_loss', {}, dict(pm=pm, lm=lm, pl=pl, ll=ll, mel_mse=a, lin_mse=b, lin_l1=c), {})


def full_size_case():
    """Full-dimension C1 (B=4, T=66, L=12) inference through the real reference with the
    build's own seeded synthetic weights (regenerated on the GPU box from the seed, so the
    125 MB of weights are never committed); outputs committed in full for mel/align and as
    a strided slice for linear."""
    import yaml
    from semi_tts_amd.synthetic import synthetic_state_dict
    cfg = yaml.safe_load(open(os.path.join(REF, 'config/semi-single-spkr-paired-data.yaml')))['model']
    dec = json.loads(json.dumps(cfg['decoder']))
    dec['decoder']['prenet_dropout'] = 0.0
    torch.manual_seed(0)
    m = RefTacotron2(80, 1025, 64, 128, dec).eval()
    sd = synthetic_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1234)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    g = torch.Generator().manual_seed(5)
    B, L, T = 4, 12, 66
    txt = torch.randn(B, L, 64, generator=g) * 0.5
    spk = torch.randn(B, 128, generator=g) * 0.5
    teacher = torch.rand(B, T, 80, generator=g)
    torch.set_num_threads(8)
    with torch.no_grad():
        mel_i, lin_i, al_i, st_i = m(txt, None, T, spk, tf_rate=0.0)
        m.decoder.query_dropout.p = 0.0   # eval mode anyway
        mel_t, lin_t, al_t, st_t = m(txt, None, teacher, spk, tf_rate=1.0)
    torch.set_num_threads(1)
    save('tts_full_c1', {}, dict(txt_embed=txt, spkr_embed=spk, teacher=teacher,
                                 mel_infer=mel_i, align_infer=al_i, stop_infer=st_i, linear_infer_s=lin_i[:, ::7, ::41],
                                 mel_tf=mel_t, align_tf=al_t, stop_tf=st_t, linear_tf_s=lin_t[:, ::7, ::41]),
         dict(seed=1234, B=B, L=L, T=T, shapes={k: list(v.shape) for k, v in m.state_dict().items()}))


def train_step_case():
    """H1: two optimisation steps of the paired TTS branch through the REAL reference classes -- VQVAE
    (codebook + speaker table + Tacotron2), util.freq_loss, optim.Optimizer, clip 5.0 -- at tiny dimensions.
    Records inputs, initial weights, dropout masks, and per step: losses, grad norm (before clipping), and after
    the first backward every parameter gradient; after the second step every parameter and buffer."""
    import yaml
    from functools import partial
    from src.optim import Optimizer as RefOptimizer
    os.chdir(REF)
    full = yaml.safe_load(open('config/semi-single-spkr-paired-data.yaml'))
    cfg = full['model']
    cfg['decoder'] = json.loads(json.dumps(TINY['paras']))
    cfg['decoder']['separate_postnet'] = True
    cfg['spkr_latent_dim'] = TINY['spkr_embed_dim']
    cfg['encoder'].update(dim=16, rnn_dim=8)               # the CTC speech encoder is built but never called here
    torch.manual_seed(11)
    m = RefVQVAE(TINY['n_mels'], TINY['linear_dim'], 43, 5, **json.loads(json.dumps(cfg)))
    os.chdir(REPO)
    g = torch.Generator().manual_seed(111)
    with torch.no_grad():
        randomize_buffers(m.tts, g)
    m.train()
    keep = lambda k: k.split('.')[0] in ('codebook', 'spkr_embed', 'tts')
    w0 = {k: v.clone() for k, v in m.state_dict().items() if keep(k)}
    B, L, T = 4, 6, 12
    text = torch.randint(3, 43, (B, L), generator=g)
    text[:, -1] = 0
    sid = torch.randint(0, 5, (B,), generator=g)
    mel = torch.rand(B, T, TINY['n_mels'], generator=g)
    linear = torch.rand(B, T, TINY['linear_dim'], generator=g)
    hp = full['hparas']
    floss = partial(ref_freq_loss, sample_rate=full['data']['audio']['sample_rate'], n_mels=TINY['n_mels'],
                    loss=hp['freq_loss_type'], differential_loss=hp['differential_loss'],
                    emphasize_linear_low=hp['emphasize_linear_low'])
    opt = RefOptimizer(m.parameters(), hp['optimizer'], hp['lr'], hp['lr_scheduler'],
                       tf_start=hp['tf_start'], tf_end=hp['tf_end'], tf_step=hp['tf_step'])
    arrays = dict(text=text, sid=sid, mel=mel, linear=linear)
    stats, n_masks = [], []
    np.random.seed(11)
    torch.manual_seed(12)
    for step in range(2):
        tf_rate = opt.pre_step(step)
        with Recorder() as rec:
            mp, lp, al, _, _, _, _, _ = m.text_to_speech(text, sid, None, None, None, None, mel, None, tf_rate)
            mel_loss, lin_loss = floss(mp, mel), floss(lp, linear)
            total = hp['tts_weight'] * (mel_loss + lin_loss)
            total.backward()
        if step == 0:
            gkeys = [k for k, p in m.named_parameters() if keep(k) and p.grad is not None]
            arrays['grad'] = [dict(m.named_parameters())[k].grad.clone() for k in gkeys]
            arrays['grad_keys'] = np.frombuffer(json.dumps(gkeys).encode(), np.uint8)
            arrays['mel_pred0'], arrays['linear_pred0'] = mp.detach().clone(), lp.detach().clone()
        gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
        opt.step()
        arrays.setdefault('mask', []).extend(rec.masks)
        n_masks.append(len(rec.masks))
        stats.append(dict(loss=float(total), mel_loss=float(mel_loss), linear_loss=float(lin_loss), grad_norm=float(gn),
                          tf_rate=float(tf_rate), lr=float(opt.opt.param_groups[0]['lr'])))
    sel = ('running_', 'codebook.learnable_table', 'spkr_embed.weight', 'tts.decoder.query_rnn.weight_hh',
           'tts.decoder.attn.v.linear.weight', 'tts.decoder.attn.loc_conv.conv.weight', 'tts.encoder.lstm.weight_hh_l0_reverse',
           'tts.postnet.0.gru.weight_hh_l0', 'tts.postnet.1.bias', 'tts.decoder.prenet.layers.0.linear.weight')
    post = {k: v for k, v in m.state_dict().items() if keep(k) and any(t in k for t in sel)}
    arrays['post'] = list(post.values())
    arrays['post_keys'] = np.frombuffer(json.dumps(list(post)).encode(), np.uint8)
    mcfg = json.loads(json.dumps(cfg))
    mcfg['codebook']['phn_attr_pth'] = ''                   # the table itself travels as w/codebook.phn_attr.weight
    save('train_step_tiny', w0, arrays, dict(stats=stats, n_masks=n_masks, model=mcfg, hparas=hp,
                                             audio=dict(sample_rate=full['data']['audio']['sample_rate'],
                                                        num_mels=TINY['n_mels'], num_freq=TINY['linear_dim']),
                                             vocab_size=43, n_spkr=5, hp=dict(TINY['paras']['decoder'], n_mels=TINY['n_mels'])))
    print(stats)


def speech_first_case():
    """The speech -> text -> speech training step of bin/train_vqvae.py:159-217,270 through the REAL reference classes at tiny
    dimensions: VQVAE.speech_to_text on [paired | unpaired] mel (CTC speech encoder -> L2 codebook with the straight-through
    estimator -> run-length merge of the unpaired part), VQVAE.text_to_speech with the unpaired latents as extra rows,
    CTC loss on the paired posteriors + freq_loss on the paired and the unpaired reconstructions, backward, clip.
    Records inputs, initial weights, dropout masks (the speech encoder's dropout is 0: nn.LSTM's inter-layer dropout cannot be
    recorded), the losses, the grad norm and EVERY parameter gradient.  Two cases: paired only, and with unpaired speech."""
    import yaml
    from functools import partial
    os.chdir(REF)
    full = yaml.safe_load(open('config/semi-single-spkr-paired-data.yaml'))
    cfg = full['model']
    cfg['decoder'] = json.loads(json.dumps(TINY['paras']))
    cfg['decoder']['separate_postnet'] = True
    cfg['spkr_latent_dim'] = TINY['spkr_embed_dim']
    cfg['encoder'].update(dim=16, rnn_dim=8, dropout=0.0)
    hp = full['hparas']
    floss = partial(ref_freq_loss, sample_rate=full['data']['audio']['sample_rate'], n_mels=TINY['n_mels'],
                    loss=hp['freq_loss_type'], differential_loss=hp['differential_loss'],
                    emphasize_linear_low=hp['emphasize_linear_low'])
    ctc = torch.nn.CTCLoss()
    EPS = 1e-10                                                  # bin/train_vqvae.py:18
    for name, with_unpaired, seed in (('speech_first_paired', False, 21), ('speech_first_unpaired', True, 22)):
        torch.manual_seed(seed)
        m = RefVQVAE(TINY['n_mels'], TINY['linear_dim'], 43, 5, **json.loads(json.dumps(cfg)))
        g = torch.Generator().manual_seed(seed + 100)
        with torch.no_grad():
            randomize_buffers(m.tts, g)
            randomize_buffers(m.asr, g)
            # keep the code usage varied at these tiny dimensions: latents comparable in size to the table rows
            m.codebook.learnable_table.mul_(0.25)
            m.asr.postnet.weight.mul_(12.0)
        m.train()
        w0 = {k: v.clone() for k, v in m.state_dict().items()}
        B, L, T = 3, 5, 24
        text = torch.randint(3, 43, (B, L), generator=g)
        text[:, -1] = 0
        text[1, -2:] = 0                                          # one shorter transcript
        sid = torch.randint(0, 5, (B,), generator=g)
        mel = torch.rand(B, T, TINY['n_mels'], generator=g)
        linear = torch.rand(B, T, TINY['linear_dim'], generator=g)
        arrays = dict(text=text, sid=sid, mel=mel, linear=linear)
        um = ul = usid = None
        if with_unpaired:
            Bu, Tu = B, 18                                        # (the reference padded_concat needs equal batch sizes)
            um = torch.rand(Bu, Tu, TINY['n_mels'], generator=g)
            ul = torch.rand(Bu, Tu, TINY['linear_dim'], generator=g)
            usid = torch.randint(0, 5, (Bu,), generator=g)
            arrays.update(unpair_mel=um, unpair_linear=ul, unpair_sid=usid)
        np.random.seed(seed)
        torch.manual_seed(seed + 1)
        with Recorder() as rec:
            pair_prob, _, unpair_prob, unpair_latent, unpair_latent_len, _, _ = m.speech_to_text(paired_mel=mel, unpaired_mel=um)
            assert (not with_unpaired) or unpair_latent is not None, 'an all-blank utterance: pick another seed'
            out = m.text_to_speech(paired_text=text, paired_sid=sid, unpaired_sid=usid, unpaired_latent=unpair_latent,
                                   unpaired_text=None, unpaired_latent_len=unpair_latent_len, paired_teacher=mel,
                                   unpaired_teacher=um, tf_rate=1.0)
            pm, pl, pa, _, upm, upl, upa, _ = out
            ctc_in = (pair_prob + EPS).transpose(0, 1).log()                                        # :432
            ctc_len = torch.LongTensor([pair_prob.shape[1]] * pair_prob.shape[0])                   # :442
            asr_loss = ctc(ctc_in, text.to_sparse().values(), ctc_len, torch.sum(text != 0, dim=-1))   # :441-444
            tts_loss = floss(pm, mel) + floss(pl, linear)
            total = hp['asr_weight'] * asr_loss + hp['tts_weight'] * tts_loss
            stats = dict(asr_loss=float(asr_loss), tts_loss=float(tts_loss))
            if with_unpaired:
                un_loss = floss(upm, um) + floss(upl, ul)                                           # :228-229
                total = total + hp['unpair_speech_weight'] * un_loss
                stats['unpair_speech_loss'] = float(un_loss)
            total.backward()
        gkeys = [k for k, p in m.named_parameters() if p.grad is not None]
        arrays['grad'] = [dict(m.named_parameters())[k].grad.clone() for k in gkeys]
        arrays['grad_keys'] = np.frombuffer(json.dumps(gkeys).encode(), np.uint8)
        arrays['pair_prob'], arrays['mel_pred'], arrays['linear_pred'] = pair_prob.detach(), pm.detach(), pl.detach()
        arrays['idx'] = pair_prob.detach().argmax(-1) if not with_unpaired else torch.cat([pair_prob, unpair_prob]).detach().argmax(-1)
        if with_unpaired:
            arrays['unpair_latent'], arrays['unpair_latent_len'] = unpair_latent.detach(), unpair_latent_len
            arrays['unpair_mel_pred'] = upm.detach()
        gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
        stats.update(loss=float(total), grad_norm=float(gn))
        arrays['mask'] = rec.masks
        mcfg = json.loads(json.dumps(cfg))
        mcfg['codebook']['phn_attr_pth'] = ''
        save(name, w0, arrays, dict(stats=stats, model=mcfg, hparas=hp,
                                    audio=dict(sample_rate=full['data']['audio']['sample_rate'], num_mels=TINY['n_mels'],
                                               num_freq=TINY['linear_dim']),
                                    vocab_size=43, n_spkr=5, hp=dict(TINY['paras']['decoder'], n_mels=TINY['n_mels']),
                                    with_unpaired=with_unpaired, n_grads=len(gkeys)))
        print(name, stats, 'grads', len(gkeys), 'masks', len(rec.masks))
    os.chdir(REPO)


def text_first_case():
    """The text -> speech -> text training step of bin/train_vqvae.py:186-205,208-224,234-250 through the REAL reference classes at tiny
    dimensions: VQVAE.text_to_speech on paired text (teacher forced) + unpaired text (rows without a teacher feed their own output
    back), the unpaired prediction detached, VQVAE.speech_to_text on [paired mel | predicted mel of the unpaired text] with
    using_fake_mel (the codebook table is detached for the fake part), CTC on the paired posteriors, freq_loss on the paired
    reconstruction, CTC on the unpaired posteriors against the unpaired text; backward, clip.  Records what speech_first_case records."""
    import yaml
    from functools import partial
    os.chdir(REF)
    full = yaml.safe_load(open('config/semi-single-spkr-paired-data.yaml'))
    cfg = full['model']
    cfg['decoder'] = json.loads(json.dumps(TINY['paras']))
    cfg['decoder']['separate_postnet'] = True
    cfg['spkr_latent_dim'] = TINY['spkr_embed_dim']
    cfg['encoder'].update(dim=16, rnn_dim=8, dropout=0.0)
    hp = dict(full['hparas'], unpair_text_weight=0.5)        # (0.0 in the shipped YAMLs: the term would not reach the gradients)
    floss = partial(ref_freq_loss, sample_rate=full['data']['audio']['sample_rate'], n_mels=TINY['n_mels'],
                    loss=hp['freq_loss_type'], differential_loss=hp['differential_loss'],
                    emphasize_linear_low=hp['emphasize_linear_low'])
    ctc = torch.nn.CTCLoss()
    EPS = 1e-10
    seed = 23
    torch.manual_seed(seed)
    m = RefVQVAE(TINY['n_mels'], TINY['linear_dim'], 43, 5, **json.loads(json.dumps(cfg)))
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        randomize_buffers(m.tts, g)
        randomize_buffers(m.asr, g)
        m.codebook.learnable_table.mul_(0.25)
        m.asr.postnet.weight.mul_(12.0)
    m.train()
    w0 = {k: v.clone() for k, v in m.state_dict().items()}
    B, L, T = 3, 5, 24
    text = torch.randint(3, 43, (B, L), generator=g)
    text[:, -1] = 0
    text[1, -2:] = 0
    sid = torch.randint(0, 5, (B,), generator=g)
    mel = torch.rand(B, T, TINY['n_mels'], generator=g)
    linear = torch.rand(B, T, TINY['linear_dim'], generator=g)
    Bu, Lu = 3, 4                                             # FRAME_PHN_RATIO * 4 = 24 frames of predicted speech, 12 CTC frames
    utext = torch.randint(3, 43, (Bu, Lu), generator=g)
    utext[:, -1] = 0
    usid = torch.randint(0, 5, (Bu,), generator=g)
    arrays = dict(text=text, sid=sid, mel=mel, linear=linear, unpair_text=utext, unpair_sid=usid)
    np.random.seed(seed)
    torch.manual_seed(seed + 1)
    with Recorder() as rec:
        pm, pl, pa, _, upm, upl, upa, _ = m.text_to_speech(paired_text=text, paired_sid=sid, unpaired_sid=usid, unpaired_latent=None,
                                                           unpaired_text=utext, unpaired_latent_len=None, paired_teacher=mel,
                                                           unpaired_teacher=None, tf_rate=1.0)
        upm = upm.detach()                                                                          # :201-202
        pair_prob, _, unpair_prob, _, _, _, _ = m.speech_to_text(paired_mel=mel, unpaired_mel=upm, using_fake_mel=True)
        ctc_in = (pair_prob + EPS).transpose(0, 1).log()
        ctc_len = torch.LongTensor([pair_prob.shape[1]] * pair_prob.shape[0])
        asr_loss = ctc(ctc_in, text.to_sparse().values(), ctc_len, torch.sum(text != 0, dim=-1))
        tts_loss = floss(pm, mel) + floss(pl, linear)
        uin = (unpair_prob + EPS).transpose(0, 1).log()                                             # :236
        ulen = torch.LongTensor([unpair_prob.shape[1]] * unpair_prob.shape[0])                      # :242
        unpair_text_loss = ctc(uin, utext.to_sparse().values(), ulen, torch.sum(utext != 0, dim=-1))   # :243-244
        total = hp['asr_weight'] * asr_loss + hp['tts_weight'] * tts_loss + hp['unpair_text_weight'] * unpair_text_loss
        stats = dict(asr_loss=float(asr_loss), tts_loss=float(tts_loss), unpair_text_loss=float(unpair_text_loss))
        total.backward()
    gkeys = [k for k, p in m.named_parameters() if p.grad is not None]
    arrays['grad'] = [dict(m.named_parameters())[k].grad.clone() for k in gkeys]
    arrays['grad_keys'] = np.frombuffer(json.dumps(gkeys).encode(), np.uint8)
    arrays['pair_prob'], arrays['unpair_prob'] = pair_prob.detach(), unpair_prob.detach()
    arrays['mel_pred'], arrays['unpair_mel_pred'] = pm.detach(), upm
    arrays['idx'] = torch.cat([pair_prob, unpair_prob]).detach().argmax(-1)
    gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
    stats.update(loss=float(total), grad_norm=float(gn))
    arrays['mask'] = rec.masks
    arrays['coins'] = np.asarray(rec.coins, np.float64)
    mcfg = json.loads(json.dumps(cfg))
    mcfg['codebook']['phn_attr_pth'] = ''
    save('text_first_unpaired', w0, arrays, dict(stats=stats, model=mcfg, hparas=hp,
                                                 audio=dict(sample_rate=full['data']['audio']['sample_rate'], num_mels=TINY['n_mels'],
                                                            num_freq=TINY['linear_dim']),
                                                 vocab_size=43, n_spkr=5, hp=dict(TINY['paras']['decoder'], n_mels=TINY['n_mels']),
                                                 n_grads=len(gkeys)))
    print('text_first_unpaired', stats, 'grads', len(gkeys), 'masks', len(rec.masks), 'unpair_prob', tuple(unpair_prob.shape))
    os.chdir(REPO)


def asr_cases():
    """CTC speech encoder (src/asr.py) at tiny dimensions: eval mode, and training mode with dropout 0 (BatchNorm batch
    statistics; the inter-layer dropout of nn.LSTM cannot be recorded, so no dropout case)."""
    from src.asr import CTC as RefCTC
    base = dict(dim=24, kernel=[3, 4, 3, 3, 3, 1], stride=[1, 2, 1, 1, 1, 1], residual=[0, 0, 1, 1, 1, 1], activation='Tanh',
                batch_norm=True, rnn_bid=True, rnn_layers=2, rnn_dim=12, layer_norm=False)
    for name, training, dropout, seed in (('asr_tiny_eval', False, 0.5, 51), ('asr_tiny_train', True, 0.0, 52)):
        torch.manual_seed(seed)
        cfg = dict(base, dropout=dropout)
        m = RefCTC(8, 10, **cfg)
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for n_, buf in m.named_buffers():
                if n_.endswith('running_mean'):
                    buf.copy_(torch.randn(buf.shape, generator=g) * 0.2)
                elif n_.endswith('running_var'):
                    buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
        m.train(training)
        w0 = {k: v.clone() for k, v in m.state_dict().items()}
        x = torch.rand(3, 14, 8, generator=g)
        with torch.no_grad():
            y = m(x)
        arrays = dict(x=x, y=y)
        if training:
            post = {k: v for k, v in m.state_dict().items() if 'running_' in k}
            arrays['post'] = list(post.values())
            arrays['post_keys'] = np.frombuffer(json.dumps(list(post)).encode(), np.uint8)
        save(name, w0, arrays, dict(cfg=cfg, in_dim=8, out_dim=10, training=training))
    # LayerNorm after the BiLSTM (layer_norm=True, src/asr.py:38-39,58): eval, and training mode with dropout 0
    for name, training, dropout, seed in (('asr_tiny_ln_eval', False, 0.5, 53), ('asr_tiny_ln_train', True, 0.0, 54)):
        torch.manual_seed(seed)
        cfg = dict(base, dropout=dropout, layer_norm=True)
        m = RefCTC(8, 10, **cfg)
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            randomize_buffers(m, g)
        m.train(training)
        w0 = {k: v.clone() for k, v in m.state_dict().items()}
        x = torch.rand(3, 14, 8, generator=g)
        with torch.no_grad():
            y = m(x)
        arrays = dict(x=x, y=y)
        if training:
            post = {k: v for k, v in m.state_dict().items() if 'running_' in k}
            arrays['post'] = list(post.values())
            arrays['post_keys'] = np.frombuffer(json.dumps(list(post)).encode(), np.uint8)
        save(name, w0, arrays, dict(cfg=cfg, in_dim=8, out_dim=10, training=training))
    # unidirectional LSTM (rnn_bid: False, src/asr.py:35-37) with LayerNorm: eval, and training mode with dropout 0 plus the
    # reference's own gradients for a recorded output gradient
    for name, training, dropout, seed in (('asr_tiny_uni_eval', False, 0.5, 57), ('asr_tiny_uni_train', True, 0.0, 58)):
        torch.manual_seed(seed)
        cfg = dict(base, dropout=dropout, layer_norm=True, rnn_bid=False)
        m = RefCTC(8, 10, **cfg)
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            randomize_buffers(m, g)
        m.train(training)
        w0 = {k: v.clone() for k, v in m.state_dict().items()}
        x = torch.rand(3, 14, 8, generator=g).requires_grad_(training)
        y = m(x)
        arrays = dict(x=x.detach(), y=y.detach())
        if training:
            dy = torch.randn(y.shape, generator=g)
            y.backward(dy)
            arrays.update(dy=dy, dx=x.grad)
            for k, p_ in m.named_parameters():
                arrays['grad/' + k] = p_.grad
            post = {k: v for k, v in m.state_dict().items() if 'running_' in k}
            arrays['post'] = list(post.values())
            arrays['post_keys'] = np.frombuffer(json.dumps(list(post)).encode(), np.uint8)
        save(name, w0, arrays, dict(cfg=cfg, in_dim=8, out_dim=10, training=training))
    # ASRPostnet (src/asr.py:67-80): eval mode (its two dropouts of 0.5 are drawn inside torch in training mode)
    from src.asr import ASRPostnet as RefPost
    torch.manual_seed(55)
    m = RefPost(12, 12).eval()
    g = torch.Generator().manual_seed(56)
    x = torch.randn(3, 9, 12, generator=g)
    with torch.no_grad():
        y = m(x)
    save('asr_postnet_tiny', {k: v.clone() for k, v in m.state_dict().items()}, dict(x=x, y=y), dict(latent_dim=12, vocab_size=12))



def pretrained_case():
    """VQVAE(pretrained_asr=..., pretrained_emb=..., pretrained_tts=...) of the REAL reference (src/vqvae.py:70-90) at tiny
    dimensions: a donor model's weights are written as two checkpoints with the key prefixes the reference expects
    ('encoder.' + speech-encoder keys [+ 'emb.weight'], the TTS model's own 'encoder.' / 'decoder.' / 'postnet.' keys), a second
    model is constructed from them, and its state_dict is recorded together with the two checkpoints.  Data only."""
    import tempfile
    import yaml
    os.chdir(REF)
    full = yaml.safe_load(open('config/semi-single-spkr-paired-data.yaml'))
    for name, bone, seed in (('pretrained_l2', 'l2', 61), ('pretrained_seperate', 'seperate', 62)):
        cfg = json.loads(json.dumps(full['model']))
        cfg['decoder'] = json.loads(json.dumps(TINY['paras']))
        cfg['decoder']['separate_postnet'] = True
        cfg['spkr_latent_dim'] = TINY['spkr_embed_dim']
        cfg['encoder'].update(dim=16, rnn_dim=8, dropout=0.0)
        cfg['codebook']['bone'] = bone
        torch.manual_seed(seed)
        donor = RefVQVAE(TINY['n_mels'], TINY['linear_dim'], 43, 5, **json.loads(json.dumps(cfg)))
        g = torch.Generator().manual_seed(seed + 100)
        with torch.no_grad():
            randomize_buffers(donor.tts, g)
            randomize_buffers(donor.asr, g)
        ck_asr = {'encoder.' + k: v.clone() for k, v in donor.asr.state_dict().items()}
        if bone == 'seperate':
            ck_asr['emb.weight'] = torch.randn(43, donor.codebook.embedding.weight.shape[1], generator=g)
        ck_tts = {k: v.clone() for k, v in donor.tts.state_dict().items()}
        with tempfile.TemporaryDirectory() as td:
            pa, pt = os.path.join(td, 'asr.pth'), os.path.join(td, 'tts.pth')
            torch.save({'model': ck_asr}, pa)
            torch.save({'model': ck_tts}, pt)
            cfg2 = json.loads(json.dumps(cfg))
            cfg2.update(pretrained_asr=pa, pretrained_tts=pt, pretrained_emb=pa if bone == 'seperate' else '')
            torch.manual_seed(seed + 1)
            m = RefVQVAE(TINY['n_mels'], TINY['linear_dim'], 43, 5, **cfg2)
        meta = dict(bone=bone, model=cfg, seed=seed + 1, flags=[bool(m.pretrain_asr), bool(m.pretrained_emb), bool(m.pretrained_tts)])
        arrays = {}
        for k, v in ck_asr.items():
            arrays['ckpt_asr/' + k] = v
        for k, v in ck_tts.items():
            arrays['ckpt_tts/' + k] = v
        save(name, {k: v for k, v in m.state_dict().items()}, arrays, meta)
    os.chdir(REPO)


def main():
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ['tts', 'vq', 'misc', 'full', 'train', 'asr', 'speech', 'text', 'variants', 'pretrained']
    if 'pretrained' in which:
        pretrained_case()
    if 'speech' in which:
        speech_first_case()
    if 'text' in which:
        text_first_case()
    if 'asr' in which:
        asr_cases()
    if 'train' in which:
        train_step_case()
    if 'variants' in which:
        # decoder variants no shipped YAML reaches (src/module.py:116-120,238-250): speaker-conditioned memory, pre-training;
        # a 2-layer text-encoder LSTM
        tts_case('tts_tiny_concat', 31, B=3, L=6, teacher=(9,), tf_rate=1.0, training=False, spkr_embed_mode='concat')
        tts_case('tts_tiny_add', 32, B=2, L=7, teacher=12, tf_rate=0.0, training=False, prenet_dropout=0.0, spkr_embed_mode='add')
        tts_case('tts_tiny_pretrain', 33, B=2, L=5, teacher=(9,), tf_rate=1.0, training=True, pretrain=True)
        tts_case('tts_tiny_enc2', 34, B=2, L=8, teacher=9, tf_rate=0.0, training=False, prenet_dropout=0.0, _enc_rnn_layer=2)
        # teacher-mean inputs (drop_dec_in), attention without location features / without the cumulative-weights channel
        tts_case('tts_tiny_dropin', 35, B=3, L=6, teacher=(18,), tf_rate=0.8, training=True, drop_dec_in=0.5)
        tts_case('tts_tiny_noloc', 36, B=2, L=7, teacher=(9,), tf_rate=1.0, training=True, loc_aware=False)
        tts_case('tts_tiny_nosum', 37, B=2, L=6, teacher=(9,), tf_rate=1.0, training=True, use_summed_weights=False)
        tts_case('tts_tiny_encdrop', 38, B=2, L=6, teacher=(9,), tf_rate=1.0, training=True, _enc_dropout=0.3)
        # normalised prenet (prenet_norm_type: the Linear wrapper's LayerNorm / BatchNorm1d, src/module.py:508-521)
        tts_case('tts_tiny_preln_infer', 41, B=3, L=7, teacher=12, tf_rate=0.0, training=False, prenet_norm_type='LayerNorm')
        tts_case('tts_tiny_preln_train', 42, B=3, L=6, teacher=(12,), tf_rate=1.0, training=True, prenet_norm_type='LayerNorm')
        tts_case('tts_tiny_prebn_infer', 43, B=3, L=7, teacher=12, tf_rate=0.0, training=False, prenet_norm_type='BatchNorm1d')
        tts_case('tts_tiny_prebn_train', 44, B=4, L=6, teacher=(12,), tf_rate=1.0, training=True, prenet_norm_type='BatchNorm1d')
        tts_case('tts_tiny_prebn_sched', 45, B=4, L=6, teacher=(12,), tf_rate=0.5, training=True, prenet_norm_type='BatchNorm1d')
        # own-output feedback through a normalised prenet: LayerNorm with scheduled sampling; BatchNorm1d whose batch of a step is
        # the rows WITHOUT a teacher only (src/module.py:197-198,205-206: prenet(mel_out[teacher_bs:]))
        tts_case('tts_tiny_preln_sched', 46, B=3, L=6, teacher=(12,), tf_rate=0.5, training=True, prenet_norm_type='LayerNorm')
        tts_case('tts_tiny_prebn_partial', 47, B=5, L=6, teacher=(9,), tf_rate=1.0, training=True, teacher_bs=2,
                 unpair_max_frame=12, prenet_norm_type='BatchNorm1d')
    if 'tts' in which:
        # eval-mode free-running inference, prenet dropout active (always-on), masks recorded
        tts_case('tts_tiny_infer', 1, B=2, L=7, teacher=15, tf_rate=0.0, training=False)
        # eval-mode, prenet dropout off: deterministic, no masks
        tts_case('tts_tiny_infer_nodrop', 2, B=3, L=9, teacher=12, tf_rate=0.0, training=False, prenet_dropout=0.0)
        # training-mode teacher forcing: BN batch stats, all three dropouts recorded
        tts_case('tts_tiny_train_tf', 3, B=4, L=6, teacher=(12,), tf_rate=1.0, training=True)
        # eval-mode teacher forcing (running-stat BN), tensor teacher
        tts_case('tts_tiny_eval_tf', 4, B=2, L=5, teacher=(9,), tf_rate=1.0, training=False)
        # scheduled sampling: coin flips recorded
        tts_case('tts_tiny_sched', 5, B=2, L=6, teacher=(12,), tf_rate=0.5, training=True)
        # partial teacher (unpaired text rows have no teacher), text-to-text cycle
        tts_case('tts_tiny_partial', 6, B=3, L=6, teacher=(9,), tf_rate=1.0, training=True, teacher_bs=2,
                 unpair_max_frame=12)
        # tensor teacher with tf_rate 0: un-divided step-count quirk (module.py:168)
        tts_case('tts_tiny_quirk', 7, B=2, L=5, teacher=(4,), tf_rate=0.0, training=False, prenet_dropout=0.0)
    if 'vq' in which:
        vq_cases()
        mean_forward_case()
    if 'misc' in which:
        postnet_class_case()
        postnet_class_train_case()
        loss_case()
    if 'full' in which:
        full_size_case()


if __name__ == '__main__':
    main()


def state_dict_keys_case():
    """key -> shape of the reference's Tacotron2 / codebooks with the shipped configuration (data only)"""
    import yaml
    os.chdir(REF)
    cfg = yaml.safe_load(open('config/semi-single-spkr-paired-data.yaml'))['model']
    torch.manual_seed(0)
    tts = RefTacotron2(80, 1025, 64, 128, json.loads(json.dumps(cfg['decoder'])))
    cb = dict(cfg['codebook'])
    cb.pop('bone')
    l2 = RefL2(43, False, **cb)
    sep = RefSep(43, False, **cb)
    os.chdir(REPO)
    out = {'tts': {k: list(v.shape) for k, v in tts.state_dict().items()},
           'l2': {k: list(v.shape) for k, v in l2.state_dict().items()},
           'seperate': {k: list(v.shape) for k, v in sep.state_dict().items()}}
    with open(os.path.join(OUT, 'state_dict_keys.json'), 'w') as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print('state_dict_keys.json', sum(len(v) for v in out.values()), 'keys')


if __name__ == '__main__' and 'keys' in sys.argv[1:]:
    state_dict_keys_case()
