#!/usr/bin/env python3
"""Debug aid (GPU box): per-tensor gradient error of one C2-size training step against float64 CPU autograd through the
oracle, worst first.  Test infrastructure (imports the oracle)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import torch   # noqa: E402
import yaml    # noqa: E402

from helpers import masks_to, split_masks   # noqa: E402
from oracle import tts_oracle as O          # noqa: E402
from oracle import vq_oracle as VQ          # noqa: E402


class Drop64(O.DropoutSource):
    def __call__(self, x, p, training):
        if (not training) or p == 0.0:
            return x
        keep = torch.full(x.shape, 1.0 - p, dtype=torch.float32)
        mk = torch.bernoulli(keep, generator=self.gen) / (1.0 - p)
        self.used.append(mk)
        return x * mk.to(x.dtype)


def main():
    from semi_tts_amd import autograd as AG
    from semi_tts_amd.synthetic import load_synthetic, synthetic_train_batch
    from semi_tts_amd.vqvae import VQVAE
    dev = torch.device('cuda:0')
    cfg = yaml.safe_load(open(os.path.join(REPO, 'config', 'semi-single-spkr-paired-data.yaml')))
    mcfg = cfg['model']
    mcfg['codebook'].update(phn_attr_pth='', proj_attr=None)
    m = VQVAE(80, 1025, 43, 109, **mcfg)
    load_synthetic(m, 321)
    m = m.to(dev).train()
    text, sid, mel, linear = synthetic_train_batch(32, 256, 3, seed=17)
    hp = dict(mcfg['decoder']['decoder'], n_mels=80, enc_dropout=0.0)
    W = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    torch.set_default_dtype(torch.float64)
    Wt = {k[4:]: (v.double() if v.is_floating_point() else v) for k, v in W.items() if k.startswith('tts.')}
    for k, v in Wt.items():
        if v.is_floating_point() and 'running_' not in k:
            v.requires_grad_()
    table, spk_table = W['codebook.learnable_table'].double().requires_grad_(), W['spkr_embed.weight'].double().requires_grad_()
    drop = Drop64('rng', generator=torch.Generator().manual_seed(23))
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    lat = VQ.l2_inference({'learnable_table': table}, text)
    enc_r = O.encoder_forward(Wt, lat, 'encoder.', True, 0.0, drop, None)
    enc_r.retain_grad()
    mel_r, al_r, _ = O.decoder_forward(Wt, enc_r, mel.double(), spk_table[sid], hp, 1.0, None, True, drop, lambda: 0.0)
    lin_r = O.postnet_forward(Wt, mel_r.detach(), True, None)
    loss_r = O.freq_loss(mel_r, mel.double(), 22050, 80) + O.freq_loss(lin_r, linear.double(), 22050, 80)
    loss_r.backward()
    torch.set_default_dtype(torch.float32)
    ref = {'tts.' + k: v.grad for k, v in Wt.items() if v.requires_grad and v.grad is not None}
    ref['codebook.learnable_table'], ref['spkr_embed.weight'] = table.grad, spk_table.grad
    masks = masks_to(split_masks(drop.used, hp, True, 1.0, 32, 32, 86, list(range(86)), hp['prenet_dim']), dev)
    hook = {}
    enc_fwd = m.tts.encoder.forward

    def enc_spy(*a, **k):
        y = enc_fwd(*a, **k)
        y.register_hook(lambda g: hook.__setitem__('denc', g.detach().clone()))
        hook['enc'] = y.detach().clone()
        return y
    m.tts.encoder.forward = enc_spy
    mel_p, lin_p, align, *_ = m.text_to_speech(text.to(dev), sid.to(dev), None, None, None, None, mel.to(dev), None, 1.0, _masks=masks)
    loss = AG.freq_loss(mel_p, mel.to(dev), 22050, 80, 'mse', True, True) + AG.freq_loss(lin_p, linear.to(dev), 22050, 80, 'mse', True, True)
    loss.backward()
    rel = lambda a, b: float((a.detach().cpu().double() - b).abs().max()) / max(float(b.abs().max()), 1e-30)
    print('loss', float(loss), float(loss_r))
    print('encoder output err %.3e   d(encoder output) relerr %.3e (scale %.3e)' %
          (rel(hook['enc'], enc_r.detach()), rel(hook['denc'], enc_r.grad), float(enc_r.grad.abs().max())))
    named = dict(m.named_parameters())
    rows = sorted(((rel(named[k].grad, g), float(g.abs().max()), k) for k, g in ref.items() if float(g.abs().max()) > 1e-12), reverse=True)
    for r in rows[:25]:
        print('%.3e  scale %.3e  %s' % r)
    # ReLU / max-pool kinks: an activation within round-off of zero can land on the other side than in the float64 run, which
    # changes ONE term of a gradient sum by 100 %.  Such an error is concentrated in a few elements: compare the largest
    # element error with the median one and with the relative L2 error of the whole tensor.
    for _, _, k in rows[:8]:
        d = (named[k].grad.detach().cpu().double() - ref[k]).abs().flatten()
        print('%-44s max %.2e  median %.2e  99.9%% %.2e  rel-L2 %.2e' % (k, float(d.max()), float(d.median()),
              float(d.kthvalue(max(1, int(0.999 * d.numel())))[0]), float(d.norm() / ref[k].norm())))


if __name__ == '__main__':
    main()
