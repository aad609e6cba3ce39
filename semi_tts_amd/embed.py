"""VQ codebook on the HIP path: mirror of the reference's src/embed.py (L2Embedding,
SeperateEmbedding) with the same constructor arguments, attributes read by callers
(`.out_dim`, `.embedding.weight`, `.create_msg()`), state_dict keys and return tuples.

forward values only: distance -> softmax -> argmax -> lookup -> straight-through value is one
fused kernel (st_vq_l2_fwd); nothing here falls back to torch arithmetic.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from . import autograd as AG

PRESERVE_INDICES = 3     # <pad>, <space>, <eos>                        ref: src/util.py:15


def read_phn_attr(path, neg_val=0):
    """Tab-separated table, first column = phoneme symbol, header row = attribute names;
    three all-zero rows are prepended for <pad>/<space>/<eos>.   ref: src/util.py:240-245"""
    rows = []
    with open(path) as f:
        next(f)                                   # header
        for line in f:
            parts = line.rstrip('\n').split('\t')
            if len(parts) > 1:
                rows.append([float(v) for v in parts[1:]])
    attr = np.asarray(rows, dtype=np.float64)
    attr[attr == 0] = neg_val
    return np.concatenate([np.zeros((PRESERVE_INDICES, attr.shape[1])), attr])


class _TableView:
    """what `codebook.embedding` returns: an object with `.weight` (the full (V, D) table)"""

    def __init__(self, weight):
        self.weight = weight


class BaseEmbedding(nn.Module):
    """ref: src/embed.py:9-55"""

    def __init__(self, vocab_size, softmax, latent_dim, commit_weight, vq_weight, temp):
        super().__init__()
        self.vocab_size = vocab_size
        self.softmax = softmax
        self.latent_dim = latent_dim
        self.out_dim = latent_dim
        self.commit_weight = commit_weight
        self.vq_weight = vq_weight
        self.ema = False
        self.phn_attr = None
        self.proj_attr = None
        self.use_phn_attr = False
        self.onehot = nn.Embedding.from_pretrained(torch.eye(vocab_size), freeze=True)
        if temp < 0:
            self.temp = nn.Parameter(torch.FloatTensor([1]))
        else:
            self.register_buffer('temp', torch.FloatTensor([temp]))

    def load_pretrained_embedding(self, old_emb):
        """ref: src/embed.py:41-48 (VQVAE's `pretrained_emb`): replace the code embedding by the checkpoint's.  Only the 'seperate'
        codebook has an `embedding` module; on the L2 codebook `embedding` is the read-only table property and the reference's
        call fails with an AttributeError -- so does this one."""
        if not isinstance(self.__dict__.get('_modules', {}).get('embedding'), nn.Embedding):
            raise AttributeError("'%s' has no embedding module to load a pretrained embedding into (as in the reference, "
                                 "src/embed.py:43: `embedding` is the table property)" % type(self).__name__)
        if 'emb.embedding.weight' in old_emb.keys():
            self.embedding = nn.Embedding.from_pretrained(old_emb['emb.embedding.weight'].data, freeze=False)
            if 'emb.temp' in old_emb.keys():
                self.temp.data = old_emb['emb.temp'].data
            if 'emb.running_tok_freq' in old_emb.keys():
                self.running_tok_freq = old_emb['emb.running_tok_freq']
            if 'emb.running_ema' in old_emb.keys():
                self.terunning_emamp = old_emb['emb.running_ema']       # (sic: the reference's attribute name, :46)
        else:
            self.embedding = nn.Embedding.from_pretrained(old_emb['emb.weight'], freeze=False)

    def _setup_attr(self, phn_attr_pth, proj_attr, latent_dim):
        self.use_phn_attr = phn_attr_pth is not None and phn_attr_pth != ''
        if self.use_phn_attr:
            assert latent_dim > proj_attr > 0, 'Currently, proj attr is necessary'
            phn_attr = torch.FloatTensor(read_phn_attr(phn_attr_pth))
            self.phn_attr = nn.Embedding.from_pretrained(phn_attr, freeze=True, padding_idx=0)
            self.proj_attr = nn.Linear(phn_attr.shape[1], proj_attr)
            return proj_attr
        return 0

    def _table_cached(self, learnable):
        """(table, its MFMA-order copy or None) of frozen-weight inference: built once per VERSION of the tensors the table is
        made of (in-place updates -- an optimiser step, load_state_dict -- bump the version counters), not on every lookup."""
        srcs = [learnable] + ([self.phn_attr.weight, self.proj_attr.weight, self.proj_attr.bias] if self.use_phn_attr else [])
        key = tuple((t.data_ptr(), t._version, str(t.device)) for t in srcs)
        ent = self.__dict__.get('_table_cache')
        if ent is None or ent[0] != key:
            table = self._table(learnable)
            packed = ops.vq_pack_table(table) if ops.vq_mfma_shape(table.shape[1], table.shape[0]) else None
            ent = (key, table, packed)
            self.__dict__['_table_cache'] = ent
        return ent[1], ent[2]

    def _table(self, learnable):
        """cat[learnable, proj_attr(phn_attr.weight)] built on device.   ref: src/embed.py:87-94,109-112"""
        if self.training and torch.is_grad_enabled():
            # differentiable: the attribute projection is an ordinary linear layer over the V table rows
            if not self.use_phn_attr:
                return learnable
            attr = AG.linear(self.phn_attr.weight, self.proj_attr.weight, self.proj_attr.bias)
            return torch.cat([learnable, attr], dim=1)
        if self.use_phn_attr:
            return ops.vq_build_table(learnable, self.phn_attr.weight, self.proj_attr.weight, self.proj_attr.bias)
        return ops.vq_build_table(learnable)

    def _lookup(self, table, idx):
        if self.training and torch.is_grad_enabled():
            return AG.gather(table, idx)
        return ops.gather_rows(table, idx)

    def create_msg(self):
        return '           | EMA update = {}\t | Temp. = {}\t| Phn. attributs = {} ( projected = {})'.format(
            self.ema, 'learnable' if type(self.temp) is nn.Parameter else self.temp.data.item(),
            self.use_phn_attr, self.proj_attr is not None)


class L2Embedding(BaseEmbedding):
    """ref: src/embed.py:57-147"""

    def __init__(self, vocab_size, ema, softmax, latent_dim, commit_weight, vq_weight, temp,
                 skip_prob, stop_grad, phn_attr_pth=None, proj_attr=None):
        super().__init__(vocab_size, softmax, latent_dim, commit_weight, vq_weight, temp)
        assert self.softmax == 'normal'
        assert not ema
        assert commit_weight == 0
        assert vq_weight == 0
        self.skip_prob = skip_prob
        self.stop_grad = stop_grad
        n_attr = self._setup_attr(phn_attr_pth, proj_attr, latent_dim)
        self.learnable_table = nn.Parameter(torch.randn((vocab_size, latent_dim - n_attr)))

    @property
    def embedding(self):
        return _TableView(self._table(self.learnable_table))

    def inference(self, txt):
        """token ids (B,L) -> vectors (B,L,latent_dim)                     ref: src/embed.py:96-103"""
        if self.training and torch.is_grad_enabled():
            return self._lookup(self._table(self.learnable_table), txt)
        return self._lookup(self._table_cached(self.learnable_table)[0], txt)

    def forward(self, enc_embs, first_n_real_mel=0):
        """enc_embs (B,S,D) -> (p_code (B,S,V), new_latent (B,S,D), 0, 0).  ref: src/embed.py:105-147.
        `first_n_real_mel` only detaches the table for part of the batch (forward values unchanged)."""
        packed = None
        if self.training and torch.is_grad_enabled():
            table = self._table(self.learnable_table)
        else:
            table, packed = self._table_cached(self.learnable_table)
        if self.training and torch.is_grad_enabled():
            # differentiable: straight-through gradient to enc_embs, p_code -> CTC gradient to enc_embs and (for the first
            # `first_n_real_mel` utterances, or all of them) to the table, scatter-add of the picked rows      :115-145
            S = enc_embs.shape[1]
            p_code, new_latent, idx = AG.vq_l2(enc_embs, table, self.temp, first_n_real_mel * S if first_n_real_mel > 0 else None,
                                               st_onehot=not self.stop_grad)                                    # :132-138
        else:
            p_code, idx, new_latent = ops.vq_l2(enc_embs.contiguous(), table, self.temp, packed=packed)
        self.last_idx = idx
        if self.training and self.skip_prob > 0 and np.random.rand() < self.skip_prob:
            new_latent = enc_embs                        # skip connection, only when training (:140-142; the draw comes after the
        return p_code, new_latent, 0, 0                  # code is picked, as in the reference)


class SeperateEmbedding(BaseEmbedding):
    """Separate ASR classifier / TTS embedding (the codebook of config/supervised.yaml).
    ref: src/embed.py:150-205"""

    def __init__(self, vocab_size, ema, softmax, latent_dim, commit_weight, vq_weight, temp,
                 skip_prob, stop_grad, phn_attr_pth=None, proj_attr=None):
        super().__init__(vocab_size, softmax, latent_dim, commit_weight, vq_weight, temp)
        assert self.softmax == 'normal'
        assert not ema
        assert commit_weight == 0
        assert vq_weight == 0
        assert skip_prob == 0
        self.stop_grad = stop_grad
        self.asr_final_layer = nn.Linear(latent_dim, vocab_size)
        n_attr = self._setup_attr(phn_attr_pth, proj_attr, latent_dim)
        self.embedding = nn.Embedding(vocab_size, latent_dim - n_attr)

    def inference(self, txt):
        return self._lookup(self._table(self.embedding.weight), txt)                 # :180-185

    def forward(self, enc_embs, first_n_real_mel=0):
        x = enc_embs.contiguous()
        if self.training and torch.is_grad_enabled():
            logits = AG.linear(x, self.asr_final_layer.weight, self.asr_final_layer.bias)
            p_code, idx = AG.softmax_argmax(logits)                                  # :190-193
            self.last_idx = idx
            table = self._table(self.embedding.weight)
            if not self.stop_grad:                                                   # ST-onehot code :198-203
                return p_code, AG.st_onehot_code(p_code, table, idx), 0, 0
            return p_code, self._lookup(table, idx), 0, 0                            # :195-197
        logits = ops.gemm(x.view(-1, x.shape[-1]), self.asr_final_layer.weight, bias=self.asr_final_layer.bias)
        p_code, idx = ops.softmax_argmax(logits.view(*x.shape[:-1], -1))             # :190-193
        self.last_idx = idx
        new_latent = ops.gather_rows(self._table(self.embedding.weight), idx)        # :195-197
        return p_code, new_latent, 0, 0
