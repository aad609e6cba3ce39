"""Deterministic synthetic weights and inputs (numpy RandomState, identical on every box).

There is no network: checkpoints and corpora are unavailable, so benchmarks and the
full-size parity fixture use random-init weights of the exact architecture.  The same
generator feeds (a) the real reference in tools/gen_golden.py and (b) the HIP modules on
the GPU box, so 125 MB of weights never have to be committed.
"""
import zlib

import numpy as np


def synthetic_state_dict(shapes, seed=1234):
    """shapes: {state_dict key: shape}.  Returns {key: float32 ndarray} (int64 for
    num_batches_tracked).  Each tensor has its own stream derived from (seed, key), so the
    result does not depend on dict order.  Scales keep activations O(1): matrices ~
    U(-a, a) with a = sqrt(3 / fan_in) * gain."""
    out = {}
    for key, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        rs = np.random.RandomState((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 31 - 1))
        if key.endswith('num_batches_tracked'):
            out[key] = np.zeros(shape, np.int64)
        elif key.endswith('running_var'):
            out[key] = rs.uniform(0.5, 1.5, shape).astype(np.float32)
        elif key.endswith('running_mean'):
            out[key] = (rs.standard_normal(shape) * 0.1).astype(np.float32)
        elif len(shape) == 1:
            if key.endswith('.weight'):                       # BatchNorm scale
                out[key] = rs.uniform(0.7, 1.3, shape).astype(np.float32)
            else:                                             # biases
                out[key] = (rs.standard_normal(shape) * 0.05).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = np.sqrt(3.0 / fan_in)
            out[key] = rs.uniform(-a, a, shape).astype(np.float32)
    return out


def load_synthetic(module, seed=1234):
    """fill a torch module's state_dict in place with synthetic_state_dict values"""
    import torch
    sd = module.state_dict()
    syn = synthetic_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed)
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(torch.from_numpy(syn[k]).to(v.device))
    return module


def synthetic_batch(B, L, T, in_dim=64, spkr_dim=128, n_mels=80, seed=5):
    """inputs of Tacotron2.forward: txt_embed (B,L,in_dim), spkr_embed (B,spkr_dim), teacher (B,T,n_mels) in [0,1)"""
    rs = np.random.RandomState(seed)
    txt = (rs.standard_normal((B, L, in_dim)) * 0.5).astype(np.float32)
    spk = (rs.standard_normal((B, spkr_dim)) * 0.5).astype(np.float32)
    mel = rs.uniform(0, 1, (B, T, n_mels)).astype(np.float32)
    return txt, spk, mel


def synthetic_train_batch(B, frames, r, vocab_size=43, n_spkr=109, n_mels=80, linear_dim=1025, seed=5):
    """one paired training batch with the shapes the reference's loader delivers:
    text ids ~ U{3..V-1} with the trailing 0 PhoneTextEncoder appends, speaker ids, mel / linear in [0,1)
    padded to a multiple of r frames the way bin/train_vqvae.py:43-46 pads (r - T % r, at least one)."""
    import torch
    rs = np.random.RandomState(seed)
    T = frames + (r - frames % r)
    L = int(np.ceil(frames / 6.0))
    text = rs.randint(3, vocab_size, (B, L)).astype(np.int64)
    text[:, -1] = 0
    sid = rs.randint(0, n_spkr, (B,)).astype(np.int64)
    mel = rs.uniform(0, 1, (B, T, n_mels)).astype(np.float32)
    linear = rs.uniform(0, 1, (B, T, linear_dim)).astype(np.float32)
    return torch.from_numpy(text), torch.from_numpy(sid), torch.from_numpy(mel), torch.from_numpy(linear)


def synthetic_cycle_batch(B, frames, r, vocab_size=43, n_spkr=109, n_mels=80, linear_dim=1025, seed=5, stretch=1.0):
    """one batch as VqvaeTrainer.fetch_data hands it to the cycles (bin/train_vqvae.py:33-53): (mel, aug_mel, linear, text, sid).
    mel / linear in [0,1) over `frames` frames, then zero-padded (SPEC_PAD_VALUE = 0) to the next multiple of r -- at least one
    frame, :43-45; aug_mel is the augmented copy the speech encoder sees: NOT padded (:47), additive noise of the configured SNR
    range and, with `stretch` != 1, a different length (time_stretch_range of config data.audio; nearest-frame resampling)."""
    import torch
    rs = np.random.RandomState(seed)
    T = frames + (r - frames % r)
    L = int(np.ceil(frames / 6.0))
    text = rs.randint(3, vocab_size, (B, L)).astype(np.int64)
    text[:, -1] = 0
    sid = rs.randint(0, n_spkr, (B,)).astype(np.int64)
    mel = np.zeros((B, T, n_mels), np.float32)
    mel[:, :frames] = rs.uniform(0, 1, (B, frames, n_mels))
    linear = np.zeros((B, T, linear_dim), np.float32)
    linear[:, :frames] = rs.uniform(0, 1, (B, frames, linear_dim))
    Ta = max(4, int(round(frames * stretch)))
    src = np.minimum((np.arange(Ta) / stretch).astype(np.int64), frames - 1)
    aug = np.clip(mel[:, src] + rs.standard_normal((B, Ta, n_mels)).astype(np.float32) * 0.05, 0.0, 1.0).astype(np.float32)
    return (torch.from_numpy(mel), torch.from_numpy(aug), torch.from_numpy(linear), torch.from_numpy(text), torch.from_numpy(sid))
