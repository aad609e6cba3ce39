"""Shared helpers for the parity tests (test infrastructure; may import the oracle)."""
import json
import os

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(REPO, 'gpurun_out', 'parity_report.jsonl')


def report(test, **vals):
    """append one line of measured errors to gpurun_out/parity_report.jsonl (best effort)"""
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, 'a') as f:
            f.write(json.dumps(dict(test=test, **{k: (float(v) if not isinstance(v, (str, int)) else v)
                                                   for k, v in vals.items()})) + '\n')
    except OSError:
        pass


def maxdiff(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


def coin_source(coins):
    it = iter(np.asarray(coins, dtype=np.float64).tolist())
    return lambda: next(it)


def split_masks(masks, hp, training, tf_rate, B, Bt, steps, step_src, P):
    """Re-arrange dropout masks recorded from the reference (in its draw order) into the
    explicit-mask dict the HIP Decoder accepts.  Draw order (src/module.py): teacher prenet
    (2), go-frame prenet (2), then per step: query dropout, decoder dropout (training only),
    prenet of the own output (2; all rows, or only the rows without a teacher)."""
    it = iter(masks)
    out = {}
    if training and hp.get('enc_dropout', 0.0) > 0:      # the encoder's conv blocks come first, (B, C, L) in the reference
        out['enc'] = [next(it).transpose(1, 2).contiguous() for _ in range(hp.get('enc_n_conv', 3))]
    pd = hp['prenet_dropout'] > 0
    if tf_rate != 0.0 and pd:
        out['teacher'] = [next(it), next(it)]
    if pd and hp.get('prenet_norm_type'):
        out['go'] = [next(it), next(it)]         # a normalised prenet maps the zero go frame to relu(beta ...): the masks matter
    elif pd:
        next(it), next(it)                       # go frame: prenet(0) == 0 whatever the mask
    own = torch.ones(steps, 2, B, P)
    q, d = [], []
    for t in range(steps):
        if training and hp['query_dropout'] > 0:
            q.append(next(it))
        if training and hp['dec_dropout'] > 0:
            d.append(next(it))
        if pd:
            if step_src[t] == -1:
                own[t, 0], own[t, 1] = next(it), next(it)
            elif Bt < B:
                own[t, 0, Bt:], own[t, 1, Bt:] = next(it), next(it)
    rest = list(it)
    assert not rest, 'unconsumed reference dropout masks: %d' % len(rest)
    if pd:
        out['own'] = own
    if q:
        out['q'] = torch.stack(q)
    if d:
        out['d'] = torch.stack(d)
    return out


def masks_to(masks, device):
    out = {}
    for k, v in masks.items():
        out[k] = [m.to(device).contiguous() for m in v] if isinstance(v, list) else v.to(device).contiguous()
    return out


def tiny_tacotron(meta, weights, device):
    """build the HIP Tacotron2 with the tiny fixture configuration and load the reference weights"""
    from semi_tts_amd.tts import Tacotron2
    cfg = meta['cfg']
    paras = json.loads(json.dumps(cfg['paras']))
    m = Tacotron2(cfg['n_mels'], cfg['linear_dim'], cfg['in_embed_dim'], cfg['spkr_embed_dim'], paras)
    m.load_state_dict(weights)
    return m.to(device)


FULL_CFG = {
    'encoder': dict(enc_n_conv=3, enc_kernel_size=5, enc_rnn_layer=1, enc_embed_dim=512, enc_dropout=0.0),
    'decoder': dict(n_frames_per_step=3, prenet_dim=256, prenet_dropout=0.5, query_rnn_dim=1024, dec_rnn_dim=1024,
                    query_dropout=0.1, dec_dropout=0.1, attn_dim=256, n_location_filters=32,
                    location_kernel_size=31, loc_aware=True, use_summed_weights=True, drop_dec_in=0.0),
    'separate_postnet': True,
}


def full_tacotron(device, seed=1234, prenet_dropout=0.0):
    """full-size Tacotron2 (the `model.decoder` section of the shipped YAMLs) with synthetic weights"""
    from semi_tts_amd.tts import Tacotron2
    from semi_tts_amd.synthetic import load_synthetic
    cfg = json.loads(json.dumps(FULL_CFG))
    cfg['decoder']['prenet_dropout'] = prenet_dropout
    m = Tacotron2(80, 1025, 64, 128, cfg)
    load_synthetic(m, seed)
    return m.to(device).eval()


def full_hp(prenet_dropout=0.0):
    hp = dict(FULL_CFG['decoder'])
    hp['prenet_dropout'] = prenet_dropout
    hp['n_mels'] = 80
    hp['enc_dropout'] = 0.0
    return hp


def tiny_vqvae(meta, weights, device, strict=False):
    """HIP VQVAE for the H1 fixture: the phoneme-attribute table comes from the fixture (data), not from a file"""
    from semi_tts_amd.vqvae import VQVAE
    cfg = json.loads(json.dumps(meta['model']))
    proj = cfg['codebook'].get('proj_attr')
    cfg['codebook']['proj_attr'] = None
    m = VQVAE(meta['audio']['num_mels'], meta['audio']['num_freq'], meta['vocab_size'], meta['n_spkr'], **cfg)
    if 'codebook.phn_attr.weight' in weights:
        cb = m.codebook
        attr = weights['codebook.phn_attr.weight']
        cb.use_phn_attr = True
        cb.phn_attr = torch.nn.Embedding.from_pretrained(attr.clone(), freeze=True, padding_idx=0)
        cb.proj_attr = torch.nn.Linear(attr.shape[1], proj)
        cb.learnable_table = torch.nn.Parameter(torch.zeros(meta['vocab_size'], cfg['codebook']['latent_dim'] - proj))
    missing = m.load_state_dict(weights, strict=False)
    assert not [k for k in missing.missing_keys if not (k.startswith('asr.') and not strict)], missing.missing_keys
    assert not (strict and missing.unexpected_keys), missing.unexpected_keys
    return m.to(device)
