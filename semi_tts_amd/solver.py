"""Solver-shaped entry points (`Solver(config, paras, mode).load_data().set_model().exec()`,
ref: main.py:65-68) for the decode path on synthetic data: the corpus, audio front end and
tokenizer of the reference are outside the hot path and absent here, so `load_data` draws seeded
synthetic batches of the shapes the real loaders would deliver.

SpecgramGenerator mirrors bin/gen_specgram.py:89-129: batched free-running decode for
`mel_len + INFERENCE_MARGIN_FRAMES` frames in eval mode, then per utterance `-mel.npy`, `-spec.npy`
(fp32) and `-align.npy` trimmed to [int(len*6)//r, text_len].
"""
import json
import os
import time

import numpy as np
import torch

from .vqvae import VQVAE, FRAME_PHN_RATIO

INFERENCE_MARGIN_FRAMES = 40        # ref: bin/gen_specgram.py:17


class BaseSolver:
    def __init__(self, config, paras, mode):
        self.config, self.paras, self.mode = config, paras, mode
        if not torch.cuda.is_available() or getattr(paras, 'cpu', False):
            raise RuntimeError('the MI355X path needs a GPU (there is no CPU fallback)')
        self.device = torch.device('cuda')
        self.exp_name = getattr(paras, 'name', None) or 'synthetic'
        self.logdir = os.path.join(getattr(paras, 'logdir', 'log/'), self.exp_name)
        self.step = 0
        audio = config['data']['audio']
        self.n_mels, self.linear_dim = audio['num_mels'], audio['num_freq']
        self.vocab_size = int(getattr(paras, 'vocab_size', 43))       # <pad>,<space>,<eos> + 40 phonemes
        self.n_spkr = int(getattr(paras, 'n_spkr', 109))              # corpus/spkr/lj_vctk.json

    def verbose(self, msg):
        if getattr(self.paras, 'verbose', True):
            print('[INFO]', msg)

    def _build_model(self):
        cfg = json.loads(json.dumps(self.config['model']))
        attr = cfg['codebook'].get('phn_attr_pth')
        if attr and not os.path.exists(attr):
            # the attribute table ships with the reference checkout, not with this repository
            self.verbose('phoneme attribute table %s not found: codebook without projected attributes' % attr)
            cfg['codebook']['phn_attr_pth'], cfg['codebook']['proj_attr'] = '', None
        return VQVAE(self.n_mels, self.linear_dim, self.vocab_size, self.n_spkr, **cfg).to(self.device)


class SpecgramGenerator(BaseSolver):
    def load_data(self):
        """synthetic test set: `n_batches` batches of (mel length only is used, text ids, speaker ids)"""
        rs = np.random.RandomState(getattr(self.paras, 'seed', 0))
        B = int(getattr(self.paras, 'batch_size', self.config['data']['corpus'].get('batch_size', 8)))
        frames = int(getattr(self.paras, 'frames', 256))
        L = int(np.ceil(frames / FRAME_PHN_RATIO))
        self.test_set = []
        for i in range(int(getattr(self.paras, 'n_batches', 1))):
            text = rs.randint(3, self.vocab_size, (B, L)).astype(np.int64)
            text[:, -1] = 0                                     # PhoneTextEncoder appends index 0 (src/text.py:65)
            sid = rs.randint(0, self.n_spkr, (B,)).astype(np.int64)
            self.test_set.append((frames, torch.from_numpy(text), torch.from_numpy(sid)))
        self.filelist = ['utt%05d' % i for i in range(B * len(self.test_set))]
        return self

    def set_model(self):
        self.model = self._build_model().eval()
        self.n_frames_per_step = self.model.n_frames_per_step
        if getattr(self.paras, 'load', None):
            ckpt = torch.load(self.paras.load, map_location=self.device)
            self.model.load_state_dict(ckpt['model'], strict=False)
            self.step = ckpt.get('global_step', 0)
        else:
            from .synthetic import load_synthetic
            load_synthetic(self.model, seed=getattr(self.paras, 'seed', 0) + 1234)
        return self

    def exec(self):
        return self.gen_specgram(self.logdir + '_%dk' % (self.step // 1000))

    def gen_specgram(self, output_dir):
        os.makedirs(output_dir, exist_ok=True)
        r = self.n_frames_per_step
        cnt, frames_out, t0 = 0, 0, time.perf_counter()
        for frames, text, sid in self.test_set:
            text, sid = text.to(self.device), sid.to(self.device)
            pad = r - frames % r                                                      # gen_specgram.py:36-37
            with torch.no_grad():
                mel, lin, align, _, _, _, _, _ = self.model.text_to_speech(
                    text, sid, None, None, None, None, frames + pad + INFERENCE_MARGIN_FRAMES, None, tf_rate=0.0)
            torch.cuda.synchronize()
            enc_step = (text != 0).sum(dim=-1).cpu().tolist()
            dec_step = [int(n * FRAME_PHN_RATIO) // r for n in enc_step]
            for i, (msp, sp, ali) in enumerate(zip(mel, lin, align)):
                name = os.path.join(output_dir, self.filelist[cnt])
                np.save(name + '-mel.npy', msp.cpu().numpy().astype(np.float32), allow_pickle=False)
                np.save(name + '-spec.npy', sp.cpu().numpy().astype(np.float32), allow_pickle=False)
                np.save(name + '-align.npy', ali[:dec_step[i], :enc_step[i]].cpu().numpy())
                cnt += 1
                frames_out += msp.shape[0]
        dt = time.perf_counter() - t0
        self.verbose('Save %d spectrograms (%d frames) in %s, %.2f s' % (cnt, frames_out, output_dir, dt))
        return cnt
