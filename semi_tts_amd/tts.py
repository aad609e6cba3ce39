"""Tacotron2 wrapper (encoder -> decoder -> CBHG mel->linear postnet) on the HIP path.
Mirror of the reference's src/tts.py:9-51: same constructor, forward signature and
state_dict keys (`encoder.*`, `decoder.*`, `postnet.0.*`, `postnet.1.*`)."""
import torch
import torch.nn as nn

from . import ops
from .module import Encoder, Decoder, CBHG


class _PostLinear(nn.Linear):
    """nn.Linear(2*n_mels, linear_dim) of src/tts.py:34 executed by the HIP GEMM"""

    def forward(self, x):
        if self.training:
            from . import autograd as AG
            return AG.conv(x, self.weight, self.bias)
        return ops.gemm(x, self.weight, bias=self.bias)


class Tacotron2(nn.Module):
    def __init__(self, n_mels, linear_dim, in_embed_dim, spkr_embed_dim, paras):
        super().__init__()
        self.n_mels = n_mels
        self.linear_dim = linear_dim
        self.separate_postnet = paras.get('separate_postnet', False)
        self.encoder = Encoder(in_embed_dim, **paras['encoder'])
        self.decoder = Decoder(n_mels, enc_embed_dim=self.encoder.enc_embed_dim, spkr_embed_dim=spkr_embed_dim,
                               **paras['decoder'])
        self.prenet_dim = self.decoder.prenet_dim
        self.prenet_dropout = self.decoder.prenet_dropout
        self.loc_aware = self.decoder.loc_aware
        self.use_summed_weights = self.decoder.use_summed_weights
        self.n_frames_per_step = self.decoder.n_frames_per_step
        self.postnet = None
        self.postnet_side = False        # training: the detached postnet branch on a second stream (the caller continues it there; TtsTrainer)
        self.postnet_stream = None
        if linear_dim is not None:
            # CBHG output size is 2 * input size                                    ref: src/tts.py:29-34
            self.postnet = nn.Sequential(CBHG(n_mels, K=8), _PostLinear(n_mels * 2, linear_dim))

    def forward(self, txt_embed, txt_lengths, teacher, spkr_embed, tf_rate=0.0, unpair_max_frame=None, _masks=None):
        """txt_embed (B,L,in_embed_dim); teacher: int (max frames, inference) or (B',T,n_mels);
        returns (mel_pred, linear_pred, alignment, stop)                          ref: src/tts.py:36-51"""
        from . import ops
        for attempt in (0, 1):
            enc_output = self.encoder(txt_embed, txt_lengths, _masks=_masks.get('enc') if _masks else None)
            try:
                mel_pred, alignment, stop = self.decoder(enc_output, txt_lengths, teacher, spkr_embed, tf_rate=tf_rate,
                                                         unpair_max_frame=unpair_max_frame, _masks=_masks)
                break
            except ops.Starved:
                # the text encoder's one-launch BiLSTM was starved of compute units (its status word is read after the decode loop of
                # an eager inference pass): the encoder output is NaN.  One launch per time step from now on, and the pass again.
                if attempt or not ops.LSTM_PERSIST or torch.is_grad_enabled():
                    raise
                ops.degrade('one-launch BiLSTM layer', 'one launch per time step (ops.LSTM_PERSIST = False)')
                ops.LSTM_PERSIST = False
        linear_pred = None
        self.postnet_stream = None
        if self.postnet is not None:
            # separate_postnet only cuts the gradient (mel_pred.detach()); forward values are identical
            if (self.postnet_side and self.separate_postnet and self.training and torch.is_grad_enabled() and mel_pred.is_cuda
                    and not ops.capturing()):
                # ... and makes the branch independent of everything behind mel_pred: CBHG forward, the linear loss and the whole CBHG
                # backward neither feed nor wait for the decoder's backward through time.  A trainer that opted in (postnet_side) gets
                # linear_pred computed on a second stream, continues the branch there (loss, backward) and joins before the optimiser.
                side = ops.side_stream(mel_pred.device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    linear_pred = self.postnet(mel_pred.detach())
                self.postnet_stream = side
            else:
                linear_pred = self.postnet(mel_pred.detach() if self.separate_postnet else mel_pred)
        return mel_pred, linear_pred, alignment, stop

    def create_msg(self):
        return ['Model spec.| Model = `TACO-2`\t| Prenet dim = {}\t| Prenet dropout = {}\t'.format(
                    self.prenet_dim, self.prenet_dropout),
                '           | Loc. aware = {}\t| frames/step = {}\t| mel2linear = {}\t| sep_post = {}\t'.format(
                    self.loc_aware, self.n_frames_per_step, self.postnet is not None, self.separate_postnet)]
