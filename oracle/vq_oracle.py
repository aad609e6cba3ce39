"""CPU oracle for the VQ codebook lookup of ttaoREtw/semi-tts (src/embed.py, src/vqvae.py).

TEST INFRASTRUCTURE ONLY (see oracle/tts_oracle.py header): only tests/, smoke() and
bench.py's cpu_baseline leg import this.  Pinned against the imported reference by
tools/gen_golden.py -> tests/golden/vq_*.npz.

Weights are a flat dict keyed like the reference's `codebook.` state_dict entries:
learnable_table (V,48|64), temp (1,), phn_attr.weight (43,31), proj_attr.{weight,bias},
and for the 'seperate' bone asr_final_layer.{weight,bias}, embedding.weight.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
Weights = Dict[str, Tensor]


def full_table(W: Weights, learnable_key: str = 'learnable_table') -> Tensor:
    """cat[learnable (V,48), proj_attr(phn_attr.weight) (V,16)] or the learnable table alone.
    ref: src/embed.py:87-94,109-112."""
    t = W[learnable_key]
    if 'proj_attr.weight' in W:
        attr = W['phn_attr.weight'].matmul(W['proj_attr.weight'].t()) + W['proj_attr.bias']
        t = torch.cat([t, attr], dim=-1)
    return t


def neg_batch_l2(x: Tensor, y: Tensor) -> Tensor:
    """-(||x||^2 + ||y||^2 - 2 x y^T), the expanded form, in the reference's association
    order ((a + b) - c).   ref: src/embed.py:208-213."""
    flat = x.reshape(-1, x.shape[-1])
    d = (flat.pow(2).sum(-1, keepdim=True) + y.pow(2).sum(-1)) - 2 * flat.matmul(y.t())
    return -d


def l2_forward(W: Weights, x: Tensor, first_n_real_mel: int = 0, stop_grad: bool = True, skip: bool = False):
    """L2Embedding.forward (stop_grad=True, skip_prob=0 in every shipped config; stop_grad=False = ST-onehot code :137-138,
    skip=True = the skip connection drawn at :140-142).
    ref: src/embed.py:105-147.
    Returns p_code (B,S,V), idx (B,S) int64, new_latent (B,S,D) with the straight-through
    value fl(fl(x + code) - x) (embed.py:145), and the table.  first_n_real_mel > 0: the similarities of the remaining
    utterances see a detached table (:115-122; same values, different gradient)."""
    B, S, D = x.shape
    table = full_table(W)
    ts = torch.relu(W['temp'])
    if first_n_real_mel > 0:
        n = first_n_real_mel
        sim = torch.cat([ts * neg_batch_l2(x[:n], table).view(n, S, -1),
                         ts * neg_batch_l2(x[n:], table.detach()).view(B - n, S, -1)], dim=0)
    else:
        sim = ts * neg_batch_l2(x, table).view(B, S, -1)                              # :124
    p = sim.softmax(dim=-1)                                                           # :127
    idx = p.argmax(dim=-1)                                                            # :130 (argmax of p, not sim)
    if stop_grad:
        code = table[idx]                                                             # :134
    else:
        onehot = torch.nn.functional.one_hot(idx, table.shape[0]).to(p.dtype)
        p_hard = p + (onehot - p).detach()                                            # :137
        code = p_hard.matmul(table)                                                   # :138  F.linear(p_hard, table.T)
    new_latent = x if skip else x + code - x.detach()                                 # :142 / :145 (straight-through estimator)
    return p, idx, new_latent, table


def l2_inference(W: Weights, txt: Tensor) -> Tensor:
    """token ids -> vectors: cat[learnable_table[txt], proj_attr(phn_attr[txt])].
    ref: src/embed.py:96-103."""
    e = W['learnable_table'][txt]
    if 'proj_attr.weight' in W:
        a = W['phn_attr.weight'][txt].matmul(W['proj_attr.weight'].t()) + W['proj_attr.bias']
        e = torch.cat([e, a], dim=-1)
    return e


def seperate_forward(W: Weights, x: Tensor, stop_grad: bool = True):
    """SeperateEmbedding.forward (stop_grad=True in config/supervised.yaml; False = the ST-onehot code, :198-203).
    ref: src/embed.py:187-205."""
    logits = x.matmul(W['asr_final_layer.weight'].t()) + W['asr_final_layer.bias']
    p = torch.softmax(logits, dim=-1)                                                 # :190
    idx = p.argmax(dim=-1)                                                            # :193
    if stop_grad:
        out = W['embedding.weight'][idx]                                              # :195
        if 'proj_attr.weight' in W:                                                   # :196-197
            a = W['phn_attr.weight'][idx].matmul(W['proj_attr.weight'].t()) + W['proj_attr.bias']
            out = torch.cat([out, a], dim=-1)
    else:
        onehot = torch.nn.functional.one_hot(idx, p.shape[-1]).to(p.dtype)
        p_hard = p + (onehot - p).detach()                                            # :199
        out = p_hard.matmul(W['embedding.weight'])                                    # :200  F.linear(p_hard, weight.T)
        if 'proj_attr.weight' in W:                                                   # :201-203
            a = p_hard.matmul(W['phn_attr.weight']).matmul(W['proj_attr.weight'].t()) + W['proj_attr.bias']
            out = torch.cat([out, a], dim=-1)
    return p, idx, out


def seperate_inference(W: Weights, txt: Tensor) -> Tensor:
    """ref: src/embed.py:180-185."""
    e = W['embedding.weight'][txt]
    if 'proj_attr.weight' in W:
        a = W['phn_attr.weight'][txt].matmul(W['proj_attr.weight'].t()) + W['proj_attr.bias']
        e = torch.cat([e, a], dim=-1)
    return e


def mean_forward(idx: np.ndarray, latent: np.ndarray, max_frames_per_phn: int
                 ) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """Run-length merge of VQ codes with blank (index 0) filtering: consecutive equal
    indices (runs capped at max_frames_per_phn+1 frames) are replaced by the mean of their
    latents; blank runs are dropped; returns None if any utterance is all-blank.
    ref: VQVAE.mean_forward, src/vqvae.py:218-257.
    idx (B,T) int, latent (B,T,D) float32 -> (padded (B,Nmax,D), lengths (B,))."""
    B, T, D = latent.shape
    seqs: List[np.ndarray] = []
    for b in range(B):
        row = idx[b].tolist()
        last_idx, last_pos = row[0], 0
        cur: List[np.ndarray] = []
        t = 0
        for t, i in enumerate(row):
            if last_idx != i or (t - last_pos) > max_frames_per_phn:                  # :231
                if last_idx != 0:
                    cur.append(latent[b, last_pos:t].mean(axis=0, dtype=np.float32))  # :234
                last_idx, last_pos = i, t
        if last_idx != 0:                                                             # :239-245
            if last_pos != T - 1:
                cur.append(latent[b, last_pos:].mean(axis=0, dtype=np.float32))
            else:
                cur.append(latent[b, t])
        if len(cur) == 0:
            return None                                                               # :248-249
        seqs.append(np.stack(cur, 0))
    n = max(s.shape[0] for s in seqs)
    out = np.zeros((B, n, D), np.float32)
    for b, s in enumerate(seqs):
        out[b, :s.shape[0]] = s
    return out, np.array([s.shape[0] for s in seqs], np.int64)


def mean_forward_torch(idx: Tensor, latent: Tensor, max_frames_per_phn: int) -> Optional[Tuple[Tensor, Tensor]]:
    """`mean_forward` above on torch tensors, differentiable with respect to `latent` (what autograd sees in the reference: slices,
    `.mean(dim=0)`, `torch.stack`, `pad_sequence`).  ref: src/vqvae.py:218-257.  idx (B,T) int64, latent (B,T,D)."""
    B, T, D = latent.shape
    seqs, lens = [], []
    for b in range(B):
        row = idx[b].tolist()
        last_idx, last_pos, cur, t = row[0], 0, [], 0
        for t, i in enumerate(row):
            if last_idx != i or (t - last_pos) > max_frames_per_phn:                  # :231
                if last_idx != 0:
                    cur.append(latent[b, last_pos:t].mean(dim=0))                     # :234
                last_idx, last_pos = i, t
        if last_idx != 0:                                                             # :239-245
            cur.append(latent[b, last_pos:].mean(dim=0) if last_pos != T - 1 else latent[b, t])
        if not cur:
            return None                                                               # :248-249
        lens.append(len(cur))
        seqs.append(torch.stack(cur, dim=0))
    return torch.nn.utils.rnn.pad_sequence(seqs, batch_first=True), torch.tensor(lens, dtype=torch.int64)
