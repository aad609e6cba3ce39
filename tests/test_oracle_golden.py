"""The CPU oracle (oracle/*.py) against golden vectors recorded from the real reference
(tools/gen_golden.py).  Runs anywhere (no GPU, no /root/reference)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import tts_oracle as O
from oracle import vq_oracle as VQ

TTS_CASES = ['tts_tiny_infer', 'tts_tiny_infer_nodrop', 'tts_tiny_train_tf', 'tts_tiny_eval_tf',
             'tts_tiny_sched', 'tts_tiny_partial', 'tts_tiny_quirk',
             # decoder / encoder variants no shipped YAML reaches: speaker-conditioned memory, pre-training, 2-layer encoder LSTM
             'tts_tiny_concat', 'tts_tiny_add', 'tts_tiny_pretrain', 'tts_tiny_enc2', 'tts_tiny_dropin', 'tts_tiny_noloc',
             'tts_tiny_nosum', 'tts_tiny_encdrop',
             # normalised prenet (prenet_norm_type LayerNorm / BatchNorm1d: eval, teacher-forced training, scheduled sampling)
             'tts_tiny_preln_infer', 'tts_tiny_preln_train', 'tts_tiny_prebn_infer', 'tts_tiny_prebn_train', 'tts_tiny_prebn_sched',
             'tts_tiny_preln_sched', 'tts_tiny_prebn_partial']


def _coin_source(coins):
    it = iter(coins.tolist())
    return lambda: next(it)


@pytest.mark.parametrize('name', TTS_CASES)
def test_tacotron2_forward_matches_reference(name):
    W, A, meta = load_golden(name)
    hp = meta['hp']
    teacher = meta['teacher'] if meta['teacher'] is not None else A['teacher']
    drop = O.DropoutSource('list', A.get('mask', []))
    stats = {}
    mel, lin, align, stop = O.tacotron2_forward(
        W, A['txt_embed'], teacher, A['spkr_embed'], hp, tf_rate=meta['tf_rate'],
        unpair_max_frame=meta['unpair_max_frame'], training=meta['training'], drop=drop,
        coin=_coin_source(A['coins']), stats_out=stats)
    assert drop.pos == len(A.get('mask', [])), 'dropout call order differs from the reference'
    assert mel.shape == A['mel'].shape and lin.shape == A['linear'].shape
    assert align.shape == A['align'].shape and stop.shape == A['stop'].shape
    # fp32, different summation order than ATen's fused kernels: 2e-5 absolute on O(1) values
    assert (mel - A['mel']).abs().max() < 2e-5
    assert (align - A['align']).abs().max() < 2e-6
    assert (stop - A['stop']).abs().max() < 2e-5
    assert (lin - A['linear']).abs().max() < 5e-5
    if meta['training']:
        import json
        keys = json.loads(bytes(A['post_keys']).decode())
        for k, v in zip(keys, A['post']):
            if k in stats:
                assert (stats[k] - v).abs().max() < 1e-5, k


def test_step_count_quirk():
    # tensor teacher + tf_rate 0 -> steps = teacher.shape[1] un-divided (module.py:168)
    W, A, meta = load_golden('tts_tiny_quirk')
    assert A['align'].shape[1] == A['teacher'].shape[1]
    assert A['mel'].shape[1] == 3 * A['teacher'].shape[1]


def test_conv_postnet_class():
    W, A, _ = load_golden('conv_postnet_tiny')
    y = O.conv_postnet_forward(W, A['x'])
    assert (y - A['y']).abs().max() < 1e-5


def test_freq_loss():
    _, A, _ = load_golden('freq_loss')
    assert abs(O.freq_loss(A['pm'], A['lm'], 22050, 80) - A['mel_mse']) < 1e-6
    assert abs(O.freq_loss(A['pl'], A['ll'], 22050, 80) - A['lin_mse']) < 1e-6
    assert abs(O.freq_loss(A['pl'], A['ll'], 22050, 80, 'l1') - A['lin_l1']) < 1e-6


def test_lr_schedule_and_padding():
    assert abs(O.lr_schedule(0, 1e-3, 'decay') - 1e-3 * 1000 ** 0.5 * 1000 ** -1.5) < 1e-12
    assert abs(O.lr_schedule(999, 1e-3, 'decay') - 1e-3) < 1e-9
    assert O.lr_schedule(5, 0.01, 'fixed') == 0.01
    assert O.padded_frames(256, 3) == 258 and O.padded_frames(64, 3) == 66 and O.padded_frames(1024, 3) == 1026
    assert O.padded_frames(255, 3) == 258      # a multiple of r gets a full extra group


@pytest.mark.parametrize('name', ['vq_l2_native', 'vq_l2_512', 'vq_l2_temp'])
def test_vq_l2(name):
    W, A, _ = load_golden(name)
    p, idx, out, table = VQ.l2_forward(W, A['x'])
    assert torch.equal(idx, A['idx']), 'VQ indices must be bit-exact'
    assert (p - A['p_code']).abs().max() < 1e-6
    assert (out - A['new_latent']).abs().max() < 1e-6
    if 'inference' in A:
        assert (VQ.l2_inference(W, A['txt']) - A['inference']).abs().max() < 1e-6
        assert (table - A['table']).abs().max() < 1e-6
    # independent statement: nearest code by direct squared distance (first index on ties)
    d = ((A['x'].double().unsqueeze(-2) - table.double()) ** 2).sum(-1)
    direct = d.argmin(-1)
    agree = (direct == idx).float().mean().item()
    assert agree > 0.95    # near-tie rows built into the fixture may differ by design


def test_vq_seperate():
    W, A, _ = load_golden('vq_seperate')
    p, idx, out = VQ.seperate_forward(W, A['x'])
    assert torch.equal(idx, A['idx'])
    assert (p - A['p_code']).abs().max() < 1e-6
    assert (out - A['new_latent']).abs().max() < 1e-6
    assert (VQ.seperate_inference(W, A['txt']) - A['inference']).abs().max() < 1e-6


def test_vq_mean_forward():
    _, A, meta = load_golden('vq_mean_forward')
    for ci in range(meta['n_cases']):
        out = VQ.mean_forward(A['idx%d' % ci].numpy(), A['lat%d' % ci].numpy(), meta['max_frames_per_phn'])
        if ('none%d' % ci) in A:
            assert out is None
            continue
        lat, ln = out
        assert np.array_equal(ln, A['len%d' % ci].numpy())
        assert np.abs(lat - A['out%d' % ci].numpy()).max() < 1e-6
        # the differentiable torch form (used by the config-size cycle tests as the autograd reference)
        lat_t, ln_t = VQ.mean_forward_torch(A['idx%d' % ci], A['lat%d' % ci], meta['max_frames_per_phn'])
        assert torch.equal(ln_t, A['len%d' % ci].long())
        assert (lat_t - A['out%d' % ci]).abs().max() < 1e-6
    assert VQ.mean_forward_torch(torch.zeros(2, 5, dtype=torch.int64), torch.ones(2, 5, 3), 3) is None


@pytest.mark.parametrize('name', ['asr_tiny_eval', 'asr_tiny_train', 'asr_tiny_ln_eval', 'asr_tiny_ln_train', 'asr_tiny_uni_eval',
                                  'asr_tiny_uni_train'])
def test_asr_oracle_against_reference(name):
    from oracle import asr_oracle as AO
    W, A, meta = load_golden(name)
    stats = {}
    y = AO.ctc_forward(W, A['x'], meta['cfg'], training=meta['training'], stats_out=stats)
    assert y.shape == A['y'].shape
    assert (y - A['y']).abs().max() < 2e-6
    if meta['training']:
        import json
        for k, v in zip(json.loads(bytes(A['post_keys']).decode()), A['post']):
            assert (stats[k] - v).abs().max() < 1e-6, k


def test_nn_module_baseline_matches_reference_recording():
    """oracle/nn_baseline.py (the cpu_baseline of bench.py: the decode loop assembled from stock nn.LSTMCell / nn.Linear /
    nn.Conv1d) reproduces what the real reference produced (free-running inference, no dropout)"""
    from oracle import nn_baseline as NB
    W, A, meta = load_golden('tts_tiny_infer_nodrop')
    hp = meta['hp']
    with torch.no_grad():
        enc = O.encoder_forward(W, A['txt_embed'])
        mel, align, stop = NB.NNDecoder(W, hp)(enc, meta['teacher'], A['spkr_embed'])
    assert (mel - A['mel']).abs().max() < 2e-5
    assert (align - A['align']).abs().max() < 1e-5
    assert (stop - A['stop']).abs().max() < 2e-5


def test_nn_module_tacotron2_assembly_against_reference_and_oracle():
    """oracle/nn_baseline.NNTacotron2 (the CPU baseline of the whole-forward and training bench lines): free-running inference
    reproduces the outputs recorded from the real reference; the teacher-forced form equals the oracle"""
    from oracle import nn_baseline as NB, tts_oracle as O
    W, A, meta = load_golden('tts_tiny_infer_nodrop')
    m = NB.NNTacotron2(W, dict(meta['hp'])).eval()
    with torch.no_grad():
        mel, lin, align, stop = m(A['txt_embed'], meta['teacher'], A['spkr_embed'])
    assert (mel - A['mel']).abs().max() < 1e-6 and (lin - A['linear']).abs().max() < 1e-5 and (align - A['align']).abs().max() < 1e-6
    W2, A2, meta2 = load_golden('tts_tiny_eval_tf')
    hp2 = dict(meta2['hp'], prenet_dropout=0.0)
    m2 = NB.NNTacotron2(W2, hp2).eval()
    with torch.no_grad():
        out = m2.forward_teacher(A2['txt_embed'], A2['teacher'], A2['spkr_embed'])
        ref = O.tacotron2_forward(W2, A2['txt_embed'], A2['teacher'], A2['spkr_embed'], hp2, tf_rate=1.0)
    assert all((a - b).abs().max() < 1e-5 for a, b in zip(out, ref))
    # one optimisation step runs and moves the weights (the bench's cpu_baseline leg of --workload train)
    m2.train()
    opt = torch.optim.Adam(m2.parameters(), lr=1e-3)
    lin_t = torch.rand_like(out[1])
    loss, gn = NB.train_step(m2, opt, A2['txt_embed'], A2['spkr_embed'], A2['teacher'], lin_t,
                             lambda p, l: O.freq_loss(p, l, 22050, hp2['n_mels']))
    assert loss == loss and gn > 0


def test_asr_postnet_oracle_against_reference():
    from oracle import asr_oracle as AO
    W, A, meta = load_golden('asr_postnet_tiny')
    y = AO.asr_postnet_forward(W, A['x'])
    assert y.shape == A['y'].shape and (y - A['y']).abs().max() < 2e-6
