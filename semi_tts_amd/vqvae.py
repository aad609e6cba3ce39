"""VQVAE orchestration on the HIP path: the caller of the hot path (H1/H2 rows of SURVEY.md 8a).

Mirrors the reference's src/vqvae.py: constructor arguments and config splatting (:26-68),
`speech_to_text` (:106-141: CTC speech encoder -> codebook -> run-length merge of the unpaired part),
`text_to_speech` with its paired / unpaired batch concatenation and output slicing (:143-207),
`mean_forward` (:218-257), `padded_concat` (:259-271).
State-dict prefixes are the reference's (`asr.*`, `codebook.*`, `spkr_embed.weight`, `tts.*`).
"""
import torch
import torch.nn as nn

from . import ops
from . import autograd as AG
from .asr import CTC, ASRPostnet
from .embed import L2Embedding, SeperateEmbedding
from .tts import Tacotron2

FRAME_PHN_RATIO = 6.0          # ref: src/vqvae.py:18
PRETRAINED_ENCODER_PREFIX = 'encoder.'      # ref: src/vqvae.py:13-15
PRETRAINED_DECODER_PREFIX = 'decoder.'
PRETRAINED_POSTNET_PREFIX = 'postnet.'


class VQVAE(nn.Module):
    def __init__(self, n_mels, linear_dim, vocab_size, n_spkr, encoder, codebook, decoder, spkr_latent_dim,
                 max_frames_per_phn, stop_threshold, asr_postnet_weight=0.0, txt_update_codebook=False,
                 pretrained_asr=None, pretrained_emb=None, pretrained_tts=None):
        super().__init__()
        self.in_dim = n_mels
        self.vocab_size = vocab_size
        self.n_spkr = n_spkr
        self.n_mels = n_mels
        self.linear_dim = linear_dim
        self.spkr_latent_dim = spkr_latent_dim
        self.stop_threshold = stop_threshold
        self.max_frames_per_phn = max_frames_per_phn
        self.txt_update_codebook = txt_update_codebook
        codebook = dict(codebook)
        self.code_bone = codebook.pop('bone')                          # :41
        self.latent_dim = codebook['latent_dim']
        self.n_frames_per_step = decoder['decoder']['n_frames_per_step']
        self.asr = CTC(n_mels, self.latent_dim, **encoder)                                    # :46
        self.time_reduce_factor = self.asr.time_reduce_factor
        self.use_asr_postnet = asr_postnet_weight > 0
        if self.use_asr_postnet:                                       # :50-52 (the reference passes latent_dim as the class count)
            self.asr_postnet_weight = asr_postnet_weight
            self.asr_postnet = ASRPostnet(self.latent_dim, self.latent_dim)
        if self.code_bone == 'l2':                                     # :56-61
            self.codebook = L2Embedding(vocab_size, False, **codebook)
        elif self.code_bone == 'seperate':
            self.codebook = SeperateEmbedding(vocab_size, False, **codebook)
        else:
            raise NotImplementedError
        self.spkr_embed = nn.Embedding(self.n_spkr, spkr_latent_dim)   # :64
        self.tts = Tacotron2(n_mels, self.linear_dim, self.codebook.out_dim, self.spkr_latent_dim, decoder)   # :68
        self._load_pretrained(pretrained_asr, pretrained_emb, pretrained_tts)

    def _load_pretrained(self, pretrained_asr, pretrained_emb, pretrained_tts):
        """Partial initialisation from earlier checkpoints with the reference's key rewriting (src/vqvae.py:70-90; advertised by
        config/*.yaml:94-97).  Every occurrence of the prefix is removed from every key (str.replace, as the reference does), the
        load is non-strict and must leave no key of the target module missing."""
        from collections import OrderedDict

        def ckpt(path):
            return torch.load(path, map_location='cpu')['model']

        def load_into(module, state, what):
            missing, _ = module.load_state_dict(state, strict=False)
            assert missing == [], 'Missing pretrained para. {} ({})'.format(missing, what)

        self.pretrain_asr = pretrained_asr is not None and pretrained_asr != ''
        if self.pretrain_asr:                                          # :71-76
            old = OrderedDict((k.replace(PRETRAINED_ENCODER_PREFIX, ''), v) for k, v in ckpt(pretrained_asr).items())
            load_into(self.asr, old, 'pretrained_asr')
        self.pretrained_emb = pretrained_emb is not None and pretrained_emb != ''
        if self.pretrained_emb:                                        # :77-80 (the reference reads the ASR checkpoint's path here)
            self.codebook.load_pretrained_embedding(ckpt(pretrained_asr))
        self.pretrained_tts = pretrained_tts is not None and pretrained_tts != ''
        if self.pretrained_tts:                                        # :81-90
            old = OrderedDict((k.replace(PRETRAINED_DECODER_PREFIX, ''), v) for k, v in ckpt(pretrained_tts).items())
            load_into(self.tts.decoder, old, 'pretrained_tts decoder')
            post = OrderedDict((k.replace(PRETRAINED_POSTNET_PREFIX, ''), v) for k, v in old.items() if PRETRAINED_POSTNET_PREFIX in k)
            load_into(self.tts.postnet, post, 'pretrained_tts postnet')

    def padded_concat(self, pair, unpair):
        """zero-pad the shorter of two (B, T, D) batches in time and stack them on the batch axis.  ref: :259-271"""
        return pair.shape[0], AG.padded_concat(pair, unpair)

    def embed_speakers(self, sid):
        if self.training and torch.is_grad_enabled():
            return AG.gather(self.spkr_embed.weight, sid)
        return ops.gather_rows(self.spkr_embed.weight, sid)

    def quantize(self, enc_latent, first_n_real_mel=0):
        """codebook lookup of speech-encoder latents: (p_code, quantized_latent)   ref: :119"""
        p_code, quantized, _, _ = self.codebook(enc_latent, first_n_real_mel)
        return p_code, quantized

    def mean_forward(self, p_code, latent):
        """run-length merge + blank filter of the quantised latents on device.      ref: :218-257"""
        return AG.mean_forward(p_code, latent, self.max_frames_per_phn)

    def speech_to_text(self, paired_mel, unpaired_mel, using_fake_mel=False, _masks=None):
        """same contract and return tuple as the reference (:106-141); `_masks` (tests only) = explicit dropout masks of the encoder"""
        use_unpaired = unpaired_mel is not None
        if use_unpaired:
            paired_mel_bs, all_mel = self.padded_concat(paired_mel, unpaired_mel)
        else:
            all_mel, paired_mel_bs = paired_mel, len(paired_mel)
        enc_latent = self.asr(all_mel, _masks) if _masks is not None else self.asr(all_mel)    # :116
        paired_post_prob = self.asr_postnet(enc_latent[:paired_mel_bs]) if self.use_asr_postnet else None   # :117
        first_n_real_mel = len(paired_mel) if using_fake_mel else 0
        p_code, quantized_latent, _, rest = self.codebook(enc_latent, first_n_real_mel)         # :119
        if use_unpaired:                                                                        # :122-133
            pair_prob, pair_latent = p_code[:paired_mel_bs], quantized_latent[:paired_mel_bs]
            unpair_prob, unpair_latent = p_code[paired_mel_bs:], quantized_latent[paired_mel_bs:]
            trim_out = self.mean_forward(unpair_prob, unpair_latent)
            unpair_latent, unpair_latent_len = trim_out if trim_out is not None else (None, None)
        else:
            pair_prob, pair_latent = p_code, quantized_latent
            unpair_prob = unpair_latent = unpair_latent_len = None
        return pair_prob, pair_latent, unpair_prob, unpair_latent, unpair_latent_len, paired_post_prob, rest

    def text_to_speech(self, paired_text, paired_sid, unpaired_sid, unpaired_latent, unpaired_text, unpaired_latent_len,
                       paired_teacher, unpaired_teacher, tf_rate, _masks=None):
        """same contract and return tuple as the reference (:143-207); `_masks` (tests only) = explicit dropout masks"""
        paired_latent = self.codebook.inference(paired_text)                                   # :147
        unpair_max_frame = None
        if unpaired_text is not None:                       # text-to-text cycle              :150-163
            assert unpaired_latent is None
            use_unpaired = True
            unpaired_latent = self.codebook.inference(unpaired_text)
            paired_latent_bs, all_latent = self.padded_concat(paired_latent, unpaired_latent)
            paired_ts = paired_teacher.shape[1]
            unpaired_ts = int(FRAME_PHN_RATIO * unpaired_text.shape[1])
            unpaired_ts += unpaired_ts % self.n_frames_per_step
            unpair_max_frame = unpaired_ts
            all_teacher = paired_teacher
            all_spkr = self.embed_speakers(torch.cat([paired_sid, unpaired_sid]))
        elif unpaired_latent is not None:                   # speech-to-speech cycle          :164-173
            use_unpaired = True
            paired_latent_bs, all_latent = self.padded_concat(paired_latent, unpaired_latent)
            paired_ts, unpaired_ts = paired_teacher.shape[1], unpaired_teacher.shape[1]
            _, all_teacher = self.padded_concat(paired_teacher, unpaired_teacher)
            all_spkr = self.embed_speakers(torch.cat([paired_sid, unpaired_sid]))       # (one lookup of both id lists = the cat of two lookups)
        else:                                                                                   # :174-180
            use_unpaired = False
            all_latent, all_teacher = paired_latent, paired_teacher
            all_spkr = self.embed_speakers(paired_sid)
        mel, linear, align, stop = self.tts(all_latent, None, all_teacher, all_spkr, tf_rate=tf_rate,
                                            unpair_max_frame=unpair_max_frame, _masks=_masks)   # :183-184
        if use_unpaired:                                                                        # :187-195
            b = paired_latent_bs
            if mel.requires_grad or linear.requires_grad:
                # (the same slices; their backward writes both gradients into one tensor instead of zero-fill + copy + add per slice)
                pm, um = AG.split_pair(mel, b, paired_ts, unpaired_ts)
                side = getattr(self.tts, 'postnet_stream', None)
                if side is not None:                 # (the postnet ran on the second stream: the backward of its split belongs there too)
                    with torch.cuda.stream(side):
                        pl, ul = AG.split_pair(linear, b, paired_ts, unpaired_ts)
                else:
                    pl, ul = AG.split_pair(linear, b, paired_ts, unpaired_ts)
            else:
                pm, um, pl, ul = mel[:b, :paired_ts], mel[b:, :unpaired_ts], linear[:b, :paired_ts], linear[b:, :unpaired_ts]
            return (pm, pl, align[:b, :paired_ts], stop[:b], um, ul, align[b:, :unpaired_ts], stop[b:])
        return mel, linear, align, stop, None, None, None, None
