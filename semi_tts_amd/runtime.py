"""hipGraph-captured decode for fixed shapes (batched inference, BASELINE config 5; also used
by bench.py): the whole autoregressive loop of Decoder.forward -- processed memory, AdaIN
statistics and `steps` x (query LSTM, query projection, attention, decoder LSTM, proj/gate,
prenet) -- is recorded once on a private HIP stream and replayed with ONE graph launch, so the
loop runs at device pace instead of host-launch pace.

Static-buffer discipline (same as any stream capture): inputs are copied into buffers owned by
this object, dropout masks are redrawn in place before each replay, outputs are views of
buffers that the next call overwrites.
"""
import torch

from . import ops


class GraphedDecoder:
    def __init__(self, decoder, B, L, frames, device, prenet_dropout=None):
        """decoder: semi_tts_amd.module.Decoder (eval mode); frames: int teacher (max frames)"""
        self.decoder = decoder
        self.frames = int(frames)
        self.steps = self.frames // decoder.n_frames_per_step
        E, S = decoder.enc_embed_dim, decoder.spkr_embed_dim
        f32 = dict(device=device, dtype=torch.float32)
        self.memory = torch.zeros(B, L, E, **f32)
        self.spkr = torch.zeros(B, S, **f32)
        p = decoder.prenet_dropout if prenet_dropout is None else prenet_dropout
        self.p = p
        self.own_mask = torch.ones(self.steps, 2, B, decoder.prenet_dim, **f32) if p > 0 else None
        self.graph = None
        self.outputs = None
        self._keep = None
        import os
        self.debug = os.environ.get('ST_CHECK_HANDOFF') == '1'    # read the hand-off status word back after every replay

    def check(self):
        """Did any replay since the last check have a starved in-launch hand-off (one device word, synchronises)?  If so the outputs
        of those replays are NaN: the loop is captured AGAIN in its two-launch form (kept for the rest of the process) and the
        last call replayed from the static input buffers with the same masks.  Returns the (possibly new) output tensors -- use
        them, not views taken before the check."""
        if ops.handoff_starved(self.decoder.handoff_status):
            ops.degrade('decode loop (query projection -> attention hand-off)', 'one launch each (attn_pq_in_fin = False)')
            self.decoder.attn_pq_in_fin = False
            self.capture()
            self.graph.launch()
            torch.cuda.synchronize()
        return self.outputs

    def _run(self):
        masks = {'own': self.own_mask} if self.own_mask is not None else None
        with torch.no_grad():
            return self.decoder(self.memory, None, self.frames, self.spkr, tf_rate=0.0, _masks=masks)

    def refresh_weights(self):
        """call after the decoder's parameters changed (load_state_dict, an optimiser step): drops the cached packed
        weights and re-captures"""
        self.decoder.__dict__.pop('_packed_cache', None)
        self.graph = None
        return self.capture()

    def capture(self):
        assert not self.decoder.training, 'graph replay is for eval-mode inference'
        # weights are frozen while a captured graph is replayed: their MFMA-order copy (75 MB read + write per packing) is
        # made once, outside the graph, instead of on every replay
        self.decoder.cache_packed = True
        # 1. eager pass inside the graph's PRIVATE memory pool: the pool now owns blocks of exactly the sizes the loop needs
        self.graph = ops.Graph()
        with self.graph.memory():
            for _ in range(2):                        # (the second pass allocates in the capture's order: weights already packed)
                self._run()
                torch.cuda.synchronize()
                self.decoder.last_tapes = None        # release them back to the pool's cache
        # 2. capture: every torch.empty below is served from that cache (no hipMalloc while capturing), and whatever is freed
        #    afterwards goes back to the pool, not to other tensors of the process -- the graph keeps writing to these addresses
        with self.graph.capture(), self.graph.memory():
            self.outputs = self._run()
        self._keep = self.decoder.last_tapes      # (the tapes stay referenced as well)
        return self

    def draw_masks(self):
        if self.own_mask is not None:
            self.own_mask.bernoulli_(1.0 - self.p).div_(1.0 - self.p)

    def __call__(self, memory=None, spkr_embed=None, redraw=True):
        if self.graph is None:
            self.capture()
        if memory is not None:
            self.memory.copy_(memory)
        if spkr_embed is not None:
            self.spkr.copy_(spkr_embed)
        if redraw:
            self.draw_masks()
        self.graph.launch()
        if self.debug:
            return self.check()
        return self.outputs


class GraphedTacotron2:
    """The whole Tacotron2.forward of free-running inference (text encoder -> decode loop -> CBHG postnet -> linear projection,
    src/tts.py:36-51) captured into ONE hipGraph for fixed shapes: ~480 launches (3 encoder convs, the BiLSTM layer as one launch, 86 x 5 decode
    launches, ~30 postnet launches) replayed with a single graph launch.  Same static-buffer discipline as GraphedDecoder."""

    def __init__(self, model, B, L, frames, device):
        self.model = model
        dec = model.decoder
        self.frames = int(frames)
        self.steps = self.frames // dec.n_frames_per_step
        f32 = dict(device=device, dtype=torch.float32)
        self.txt = torch.zeros(B, L, model.encoder.convs[0][0].conv.in_channels, **f32)
        self.spkr = torch.zeros(B, dec.spkr_embed_dim, **f32)
        self.p = dec.prenet_dropout
        self.own_mask = torch.ones(self.steps, 2, B, dec.prenet_dim, **f32) if self.p > 0 else None
        self.graph = None
        self.outputs = None
        self._keep = None

    def _run(self):
        masks = {'own': self.own_mask} if self.own_mask is not None else None
        with torch.no_grad():
            return self.model(self.txt, None, self.frames, self.spkr, tf_rate=0.0, _masks=masks)

    def capture(self):
        assert not self.model.training, 'graph replay is for eval-mode inference'
        self.model.decoder.cache_packed = True
        # (as GraphedDecoder.capture: warm-up and capture inside the graph's private memory pool -- the encoder / postnet
        # intermediates and GEMM workspaces allocated by the forward are freed after the capture, and the replay still uses them)
        self.graph = ops.Graph()
        with self.graph.memory():
            for _ in range(2):                        # warms the pool, the packed weights, the tap-major conv weights; the second
                self._run()                           # pass allocates in the capture's order
                torch.cuda.synchronize()
                self.model.decoder.last_tapes = None
        with self.graph.capture(), self.graph.memory():
            self.outputs = self._run()
        self._keep = self.model.decoder.last_tapes
        return self

    def draw_masks(self):
        if self.own_mask is not None:
            self.own_mask.bernoulli_(1.0 - self.p).div_(1.0 - self.p)

    def check(self):
        """as GraphedDecoder.check, for the decode loop's hand-off AND the text encoder's one-launch BiLSTM: a starved launch means
        NaN outputs; the forward is captured again in the multi-launch form(s) and the last call replayed.  Returns the outputs."""
        bad_h = ops.handoff_starved(self.model.decoder.handoff_status)
        bad_p = ops.persist_starved()
        if bad_h:
            ops.degrade('decode loop (query projection -> attention hand-off)', 'one launch each (attn_pq_in_fin = False)')
            self.model.decoder.attn_pq_in_fin = False
        if bad_p:
            ops.degrade('one-launch BiLSTM layer', 'one launch per time step (ops.LSTM_PERSIST = False)')
            ops.LSTM_PERSIST = False
        if bad_h or bad_p:
            self.capture()
            self.graph.launch()
            torch.cuda.synchronize()
        return self.outputs

    def __call__(self, txt_embed=None, spkr_embed=None, redraw=True):
        if self.graph is None:
            self.capture()
        if txt_embed is not None:
            self.txt.copy_(txt_embed)
        if spkr_embed is not None:
            self.spkr.copy_(spkr_embed)
        if redraw:
            self.draw_masks()
        self.graph.launch()
        return self.outputs


class GraphedSpeechToText:
    """VQVAE.speech_to_text of evaluation (src/vqvae.py:106-141: CTC speech encoder -> codebook lookup -> run-length merge of the
    unpaired part) captured into ONE hipGraph for fixed shapes: the ~25 launches of the speech encoder (6 conv layers, 2 BiLSTM
    layers step by step), the nearest-code search on the cached MFMA-order table and the segmented-mean kernel replay with a
    single graph launch.  The only host work left is the reference's own data-dependent decision (an all-blank utterance drops
    the unpaired part; the longest merged sequence fixes the output length): one small device -> host copy of the lengths AFTER
    the replay.  Same static-buffer discipline as GraphedDecoder."""

    def __init__(self, model, B_pair, T, device, B_unpair=0):
        # (the ASRPostnet output of speech_to_text is not part of the captured graph: the tuple's pair_post_prob slot would be wrong)
        assert not getattr(model, 'use_asr_postnet', False), 'GraphedSpeechToText: use_asr_postnet is not captured; call model.speech_to_text'
        self.model = model
        self.Bp, self.Bu = int(B_pair), int(B_unpair)
        self.mel = torch.zeros(self.Bp + self.Bu, int(T), model.n_mels, device=device, dtype=torch.float32)
        self.graph = None
        self.outputs = None

    def _run(self):
        from . import autograd as AG
        m = self.model
        with torch.no_grad():
            enc = m.asr(self.mel)
            p_code, quantized, _, _ = m.codebook(enc, 0)
            merged = None
            if self.Bu > 0:
                merged = AG._MeanForwardFn.apply(p_code[self.Bp:].contiguous(), quantized[self.Bp:].contiguous(), m.max_frames_per_phn)
        return p_code, quantized, merged

    def capture(self):
        assert not self.model.training, 'graph replay is for eval-mode inference'
        self.graph = ops.Graph()
        with self.graph.memory():
            for _ in range(2):
                self._run()
                torch.cuda.synchronize()
        with self.graph.capture(), self.graph.memory():
            self.outputs = self._run()
        return self

    def check(self):
        """a one-launch LSTM layer of a replay since the last check starved of compute units (synchronises)?  Then the graph is
        captured again with one launch per time step and replayed on the inputs still in the static buffer; returns True if so"""
        if not ops.persist_starved():
            return False
        ops.degrade('one-launch BiLSTM layer', 'one launch per time step (ops.LSTM_PERSIST = False)')
        ops.LSTM_PERSIST = False
        self.capture()
        self.graph.launch()
        torch.cuda.synchronize()
        return True

    def __call__(self, paired_mel=None, unpaired_mel=None):
        """-> the tuple VQVAE.speech_to_text returns (views of buffers the next call overwrites)"""
        if self.graph is None:
            self.capture()
        if paired_mel is not None:
            self.mel[:self.Bp, paired_mel.shape[1]:].zero_()        # zero padding past a shorter input, as padded_concat / the eager path do
            self.mel[:self.Bp, :paired_mel.shape[1]].copy_(paired_mel)
        if unpaired_mel is not None:
            self.mel[self.Bp:].zero_()
            self.mel[self.Bp:, :unpaired_mel.shape[1]].copy_(unpaired_mel)
        self.graph.launch()
        p_code, quantized, merged = self.outputs
        if self.Bu == 0:
            return p_code, quantized, None, None, None, None, 0
        out, lens = merged
        lens_h = lens.cpu()
        if int(lens_h.min()) == 0:                      # an all-blank utterance: the unpaired speech is ignored (:129-133)
            return p_code[:self.Bp], quantized[:self.Bp], p_code[self.Bp:], None, None, None, 0
        return (p_code[:self.Bp], quantized[:self.Bp], p_code[self.Bp:], out[:, :int(lens_h.max())], lens.long(), None, 0)
