"""Solver-shaped entry points (`Solver(config, paras, mode).load_data().set_model().exec()`,
ref: main.py:65-68) for the decode path on synthetic data: the corpus, audio front end and
tokenizer of the reference are outside the hot path and absent here, so `load_data` draws seeded
synthetic batches of the shapes the real loaders would deliver.

SpecgramGenerator mirrors bin/gen_specgram.py:89-129: batched free-running decode for
`mel_len + INFERENCE_MARGIN_FRAMES` frames in eval mode, then per utterance `-mel.npy`, `-spec.npy`
(fp32) and `-align.npy` trimmed to [int(len*6)//r, text_len].
"""
import json
import math
import os
import time

import numpy as np
import torch

from . import ops
from .vqvae import VQVAE, FRAME_PHN_RATIO

INFERENCE_MARGIN_FRAMES = 40        # ref: bin/gen_specgram.py:17


class BaseSolver:
    def __init__(self, config, paras, mode):
        self.config, self.paras, self.mode = config, paras, mode
        if not torch.cuda.is_available() or getattr(paras, 'cpu', False):
            raise RuntimeError('the MI355X path needs a GPU (there is no CPU fallback)')
        self.device = torch.device('cuda', torch.cuda.current_device())
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

    # -- checkpoints: the reference's on-disk format {"model", "optimizer", "global_step"} (src/solver.py:118-135, :203-216)
    def save_checkpoint(self, f_name, score=0.0):
        ckpdir = os.path.join(getattr(self.paras, 'ckpdir', 'ckpt/'), self.exp_name)
        os.makedirs(ckpdir, exist_ok=True)
        path = os.path.join(ckpdir, f_name)
        full = {'model': self.model.state_dict(), 'global_step': self.step}
        if getattr(self, 'optimizer', None) is not None:
            full['optimizer'] = self.optimizer.get_opt_state_dict()
        torch.save(full, path)
        self.verbose('Saved checkpoint (step = %d, score = %.2f) and status @ %s' % (self.step, score, path))
        return path

    def load_ckpt(self):
        """load `paras.load` into self.model (and, in training mode, the optimizer state and step counter)"""
        path = getattr(self.paras, 'load', None)
        if not path:
            return False
        ckpt = torch.load(path, map_location=self.device)
        # strict, like the reference (bin/gen_specgram.py:80, bin/train_vqvae.py:106): a checkpoint of another architecture
        # must not leave the constructor's random weights in place silently
        self.model.load_state_dict(ckpt['model'], strict=True)
        if self.mode == 'train':
            self.step = ckpt.get('global_step', 0)
            if getattr(self, 'optimizer', None) is not None and 'optimizer' in ckpt:
                self.optimizer.load_opt_state_dict(ckpt['optimizer'])
            self.verbose('Load ckpt from %s, restarting at step %d' % (path, self.step))
        else:
            self.step = ckpt.get('global_step', 0)
            self.verbose('Evaluation target = %s (step %d)' % (path, self.step))
        return True

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
        # utterance-sharded replicas: under torch.distributed every rank decodes and writes only its own
        # contiguous range of each batch (no collective, no two ranks writing the same file)
        from . import parallel
        self.rank, self.world = parallel.rank_world()
        return self

    def set_model(self):
        self.model = self._build_model().eval()
        self.n_frames_per_step = self.model.n_frames_per_step
        if not self.load_ckpt():
            from .synthetic import load_synthetic
            load_synthetic(self.model, seed=getattr(self.paras, 'seed', 0) + 1234)
        return self

    def exec(self):
        return self.gen_specgram(self.logdir + '_%dk' % (self.step // 1000))

    def gen_specgram(self, output_dir):
        os.makedirs(output_dir, exist_ok=True)
        r = self.n_frames_per_step
        cnt, frames_out, t0 = 0, 0, time.perf_counter()
        rank, world = getattr(self, 'rank', 0), getattr(self, 'world', 1)
        from .parallel import shard_range
        next_first = 0
        for frames, text, sid in self.test_set:
            batch_first, next_first = next_first, next_first + text.shape[0]
            lo, hi = shard_range(text.shape[0], world, rank)
            if hi <= lo:
                continue
            names = self.filelist[batch_first + lo:batch_first + hi]
            text, sid = text[lo:hi].to(self.device), sid[lo:hi].to(self.device)
            pad = r - frames % r                                                      # gen_specgram.py:36-37
            with torch.no_grad():
                mel, lin, align, _, _, _, _, _ = self.model.text_to_speech(
                    text, sid, None, None, None, None, frames + pad + INFERENCE_MARGIN_FRAMES, None, tf_rate=0.0)
            torch.cuda.synchronize()
            # (the eager forward has already checked the decode loop's hand-off status word: ops.check_handoff)
            ops.check_persist_status(self.device)
            if not (bool(torch.isfinite(mel).all()) and bool(torch.isfinite(lin).all())):
                raise RuntimeError('gen_specgram: non-finite spectrogram for batch starting at %s -- nothing written' % names[0])
            enc_step = (text != 0).sum(dim=-1).cpu().tolist()
            dec_step = [int(n * FRAME_PHN_RATIO) // r for n in enc_step]
            for i, (msp, sp, ali) in enumerate(zip(mel, lin, align)):
                name = os.path.join(output_dir, names[i])
                np.save(name + '-mel.npy', msp.cpu().numpy().astype(np.float32), allow_pickle=False)
                np.save(name + '-spec.npy', sp.cpu().numpy().astype(np.float32), allow_pickle=False)
                np.save(name + '-align.npy', ali[:dec_step[i], :enc_step[i]].cpu().numpy())
                cnt += 1
                frames_out += msp.shape[0]
        dt = time.perf_counter() - t0
        if rank == 0 or world == 1:
            self.verbose('Save %d spectrograms%s (%d frames) in %s, %.2f s' %
                         (cnt, ' on rank 0 of %d' % world if world > 1 else '', frames_out, output_dir, dt))
        return cnt


class LazyStats(dict):
    """step statistics whose device scalars become Python floats when they are READ (st['loss'], st.items(), ...): a training loop
    that only logs every n-th step never waits for the GPU in between (TtsTrainer.async_stats)"""

    on_nonfinite = None      # callable(stats): runs once when a non-finite 'grad_norm' is read (the device skipped that update)

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if torch.is_tensor(v):
            v = float(v)
            dict.__setitem__(self, k, v)
            if k == 'grad_norm' and not math.isfinite(v) and self.on_nonfinite is not None:     # (NaN or inf: what the device guard skips)
                cb, self.on_nonfinite = self.on_nonfinite, None
                cb(self)
        return v

    def put(self, k, v):
        """the value of device scalar k, read by somebody else (drain_stats reads many in one copy); same side effects as reading it here"""
        dict.__setitem__(self, k, v)
        if k == 'grad_norm' and not math.isfinite(v) and self.on_nonfinite is not None:
            cb, self.on_nonfinite = self.on_nonfinite, None
            cb(self)

    def materialise(self):
        """read every device scalar (one wait for the GPU): the entry no longer holds device memory"""
        for k in dict.keys(self):
            self[k]
        return self

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return [(k, self[k]) for k in dict.keys(self)]

    def values(self):
        return [self[k] for k in dict.keys(self)]


class TtsTrainer(BaseSolver):
    """Synthetic-batch counterpart of the paired TTS branch of VqvaeTrainer.exec (bin/train_vqvae.py:132-270)
    with BaseSolver.backward (src/solver.py:138-151): per step

        tf_rate = optimizer.pre_step(step)                       (lr schedule, zero_grad)
        mel, linear = model.text_to_speech(text, sid, ..., paired_teacher=mel, tf_rate)
        loss = tts_weight * (freq_loss(mel) + freq_loss(linear))
        loss.backward(); [all-reduce over ranks]; clip_grad_norm_(5.0); optimizer.step()

    Forward, loss, backward, gradient clipping and Adam all run on the HIP kernels (semi_tts_amd/autograd.py,
    semi_tts_amd/optim.py).  The paired TTS branch alone (`main.py --tts-only`, `bench.py --workload train`); the whole loop of
    the reference -- the alternating speech <-> text cycles with the CTC speech encoder, codebook and run-length merge -- is
    VqvaeTrainer below, main.py's default mode."""
    GRAD_CLIP = 5.0
    STATIC_GRAPH = True      # the paired TTS step produces the same gradients in the same order every step (GradReducer: one hook per bucket)
    async_stats = False      # True: train_step never waits for the GPU -- LazyStats, NaN steps skipped on the device (optim.FusedAdam guard)

    @staticmethod
    def clip_grad_norm_(params, max_norm, pre_scale=1.0):
        from .optim import clip_grad_norm_
        return clip_grad_norm_(list(params), max_norm, pre_scale)

    def _clip(self):
        """clip_grad_norm_(GRAD_CLIP) over the model; under data parallelism the 1 / world of the gradient average rides in it"""
        scale = getattr(self, '_grad_scale', 1.0)
        params = self.__dict__.get('_clip_params')
        if params is None:       # (nn.Module.parameters() walks the module tree through three generators: 0.25 ms per call for ~110 parameters)
            params = self._clip_params = list(self.model.parameters())
        if scale != 1.0:
            return self.clip_grad_norm_(params, self.GRAD_CLIP, pre_scale=scale)
        return self.clip_grad_norm_(params, self.GRAD_CLIP)

    def __init__(self, config, paras, mode='train'):
        super().__init__(config, paras, mode)
        hp = config['hparas']
        self.hp = hp
        self.max_step = int(getattr(paras, 'max_step', None) or hp.get('max_step', 1))
        self.tts_weight = float(hp.get('tts_weight', 1.0))
        self.sample_rate = config['data']['audio']['sample_rate']
        self.log = []
        self._unread = []
        if getattr(paras, 'async_stats', False):
            self.async_stats = True

    def load_data(self):
        from .synthetic import synthetic_train_batch
        B = int(getattr(self.paras, 'batch_size', None) or self.config['data']['corpus'].get('batch_size', 8))
        frames = int(getattr(self.paras, 'frames', 256))
        n = int(getattr(self.paras, 'n_batches', 1))
        self.r = self.config['model']['decoder']['decoder']['n_frames_per_step']
        rank = int(os.environ.get('RANK', 0))
        self.batches = [synthetic_train_batch(B, frames, self.r, self.vocab_size, self.n_spkr, self.n_mels, self.linear_dim,
                                              seed=1000 * rank + i + getattr(self.paras, 'seed', 0))
                        for i in range(n)]
        return self

    def set_model(self):
        from .optim import Optimizer
        from .synthetic import load_synthetic
        self.model = self._build_model().train()
        self._clip_params = None
        hp = self.hp
        self.optimizer = Optimizer(self.model.parameters(), hp['optimizer'], hp['lr'], hp['lr_scheduler'],
                                   **{k: hp[k] for k in ('tf_start', 'tf_end', 'tf_step') if k in hp})
        if not self.load_ckpt():
            load_synthetic(self.model, seed=getattr(self.paras, 'seed', 0) + 1234)
        from . import parallel
        parallel.sync_batchnorm(True)        # no-op for a single process; global-batch statistics under torch.distributed
        parallel.broadcast_parameters(self.model)
        self._attach_reducer()
        # the detached postnet branch on a second stream (Tacotron2.postnet_side; ST_POSTNET_SIDE=0 turns it off) -- without data parallelism
        # only: a gradient bucket that goes out while the backward still runs must not mix gradients of two streams.  Bitwise the serial step
        # (test_postnet_branch_on_a_second_stream_gives_bitwise_the_serial_step).  Measured (round 6, A/B pairs in one process on the box of
        # the final artifacts): 8.83 -> 8.62 ms per C2 training step, with or without the loop graphs (earlier boxes of the round: -0.07 ...
        # -0.18 ms): the ~1 ms of CBHG forward / backward mostly hides behind the decoder's backward through time, the host's issue time
        # (7.9 ms) is the next floor.
        self.postnet_side = (self.reducer is None and bool(getattr(self.model.tts, 'separate_postnet', False))
                             and os.environ.get('ST_POSTNET_SIDE', '1') != '0')
        # (Tacotron2.postnet_side itself is raised only around the trainer's own text_to_speech calls, _tts: anybody else who calls the model
        # gets every output on the stream it called on)
        return self

    def _tts(self, *args, **kw):
        """VQVAE.text_to_speech for a step of THIS trainer: with `postnet_side` the postnet output comes back on the second stream
        (Tacotron2.postnet_stream) and the caller continues the branch there (train_step, VqvaeTrainer._side_branch)"""
        tts = self.model.tts
        tts.postnet_side = bool(getattr(self, 'postnet_side', False))
        try:
            return self.model.text_to_speech(*args, **kw)
        finally:
            tts.postnet_side = False

    def _attach_reducer(self):
        """under torch.distributed: gradients live in flat buckets that are all-reduced while the backward pass still runs"""
        from . import parallel
        self.reducer = None
        if parallel.dist_on():               # more than one rank, or collectives forced in a world of one (bench.py --dist)
            # the static-graph promise (one hook per bucket) only where the step's autograd graph cannot change: no teacher-forcing
            # schedule (own-output feedback adds the prenet's gradients to the graph); the reducer checks the promise anyway
            static = self.STATIC_GRAPH and not getattr(self.optimizer, 'tf_type', False)
            self.reducer = parallel.GradReducer(self.model.parameters(), static_graph=static, defer_average=True)
        return self.reducer

    def _reduce_gradients(self):
        """sum the gradients over the ranks; what they must still be multiplied by (1 / world) goes into the clip launch"""
        from . import parallel
        self._grad_scale = 1.0
        if getattr(self, 'reducer', None) is not None:
            n = self.reducer.finish()
            self._grad_scale = self.reducer.grad_scale
            return n
        return parallel.allreduce_gradients(self.model.parameters())

    def freq_loss(self, pred, label):
        from . import autograd as AG
        hp = self.hp
        return AG.freq_loss(pred, label, self.sample_rate, self.n_mels, hp.get('freq_loss_type', 'mse'),
                            hp.get('differential_loss', True), hp.get('emphasize_linear_low', True))

    def train_step(self, text, sid, mel, linear, _masks=None):
        from . import parallel
        parallel.collective_counts(reset=True)
        tf_rate = self.optimizer.pre_step(self.step)
        if getattr(self, 'reducer', None) is not None:
            self.reducer.prepare()
        mel_pred, linear_pred, align, _, _, _, _, _ = self._tts(
            text, sid, None, None, None, None, mel, None, tf_rate, _masks=_masks)
        from . import autograd as AG
        side = getattr(self.model.tts, 'postnet_stream', None)
        w = self.tts_weight
        if side is not None:
            # separate_postnet (src/tts.py:47-50): the postnet saw mel_pred.detach(), on a second stream.  Its loss and its whole backward
            # (CBHG incl. both GRU passes: ~1 ms that neither feeds nor waits for the decoder's backward through time) stay on that
            # stream, beside the main chain; the streams join before the gradients are used.  d total / d linear_loss = tts_weight.
            with torch.cuda.stream(side):
                linear_loss = self.freq_loss(linear_pred, linear)
                lin_total, = AG.scalar_combine([[w]], [linear_loss])
                lin_total.backward()
                ops.flush_wgrads()
                ev = self.__dict__.setdefault('_side_event', torch.cuda.Event())
                ev.record(side)
            ops.side_pending(ev)                      # (the one-launch BiLSTM backward of the text encoder waits for it: it needs every compute unit)
            mel_loss = self.freq_loss(mel_pred, mel)
            mel_total, = AG.scalar_combine([[w]], [mel_loss])
            mel_total.backward()
            torch.cuda.current_stream().wait_event(ev)
            ops.side_pending(None)
            with torch.no_grad():
                total, = AG.scalar_combine([[1.0, 1.0]], [mel_total.detach(), lin_total.detach()])
        else:
            mel_loss = self.freq_loss(mel_pred, mel)
            linear_loss = self.freq_loss(linear_pred, linear)
            total, = AG.scalar_combine([[w, w]], [mel_loss, linear_loss])     # (one launch; `w * (a + b)` is two, and three backward)
            total.backward()
        self._reduce_gradients()
        grad_norm = self._clip()
        from .optim import FusedAdam
        if self.async_stats and torch.is_tensor(grad_norm) and grad_norm.is_cuda and isinstance(getattr(self.optimizer, 'opt', None), FusedAdam):
            # no host round trip inside the step: the NaN check of BaseSolver.backward (src/solver.py:147-150) runs on the device (a
            # non-finite norm makes the Adam launch a no-op) and the statistics stay device scalars until somebody reads them
            opt_step = getattr(getattr(self.optimizer, 'opt', None), 'guarded_steps', 0)      # (the key of this update in the optimiser's log)
            self.optimizer.step(guard_norm=grad_norm)
            self.step += 1
            st = self._ring_stats(dict(loss=total, mel_loss=mel_loss, linear_loss=linear_loss, grad_norm=grad_norm),
                                  tf_rate=tf_rate, lr=self.optimizer.lr_at(self.step - 1), step=self.step - 1, opt_step=opt_step)
            return st
        gn = float(grad_norm)
        if gn != gn:
            self.check_device_status()                   # (a starved one-launch LSTM layer is an error, not a skipped step)
            self.verbose('Error : grad norm is NaN @ step ' + str(self.step))
        else:
            self.optimizer.step()
        self.step += 1
        return dict(loss=float(total.detach()), mel_loss=float(mel_loss.detach()), linear_loss=float(linear_loss.detach()), grad_norm=gn,
                    tf_rate=tf_rate, lr=self.optimizer.lr_at(self.step - 1))

    STATS_COLS = 8           # device scalars per row of the statistics ring

    def _ring_stats(self, scalars, **extras):
        """async_stats: the step's device scalars go into a row of a persistent ring: LazyStats then holds VIEWS, no allocation outlives
        the step, and the caching allocator hands every tensor of the next step the address it had in this one -- which is what
        lets the optimiser's device-side tables (optim.hip: mt_table) be reused instead of rebuilt (scalars kept alive for 4 ... 8 steps
        made the addresses wander with a period of ~13 steps).  An event behind the row lets drain_stats wait for THIS step only."""
        keys = tuple(scalars)
        vals = [scalars[k] for k in keys]
        dev = vals[-1].device
        ring = self.__dict__.get('_stats_ring')
        if ring is None or ring.device != dev:
            ring = self._stats_ring = torch.zeros(self.STATS_RING, self.STATS_COLS, device=dev)
            self._stats_slot = 0
            self._stats_host = torch.zeros(self.STATS_RING, self.STATS_COLS).pin_memory()
            self._stats_events = [torch.cuda.Event() for _ in range(self.STATS_RING)]
            self._stats_stream = torch.cuda.Stream(device=dev)
        slot = self._stats_slot
        row = ring[slot]
        self._stats_slot = (slot + 1) % self.STATS_RING
        torch.stack([v.detach().reshape(()) for v in vals], out=row[:len(keys)])
        self._stats_events[slot].record()                 # (drain_stats waits for THIS step only, on a stream of its own)
        st = LazyStats({k: row[i] for i, k in enumerate(keys)}, **extras)
        st.ring_slot, st.ring_keys = slot, keys
        st.on_nonfinite = self._skipped_on_device
        self._unread.append(st)
        # bounded: at most STATS_WINDOW steps of device scalars are alive.  While the one-launch BiLSTM is in use the window is short:
        # a starved layer poisons every forward until somebody looks, and every poisoned step is an update skipped on the device
        # (only the older half is read: the wait is for a step that finished a while ago, the GPU keeps the newer half queued and the
        # host its lead -- a full drain every eighth step cost the data-parallel step 0.6 ms once the host was no longer far ahead)
        window = min(self.STATS_WINDOW_PERSIST if ops.LSTM_PERSIST else self.STATS_WINDOW, self.STATS_RING - 1)
        if len(self._unread) >= window:
            self.drain_stats(keep=window // 2)
        return st

    STATS_WINDOW = 64        # async_stats: steps whose statistics may stay unread on the device
    STATS_RING = 256         # rows of the ring the unread statistics live in (> any window)
    STATS_WINDOW_PERSIST = int(os.environ.get('ST_STATS_WINDOW_PERSIST', '8'))     # ... while ops.LSTM_PERSIST is on (see train_step)

    def check_device_status(self):
        """a kernel of the training step reported starvation since the last check (the one-launch BiLSTM's status word)?  Such a
        forward is NaN and the guarded Adam skips its update: the layer runs as one launch per time step from now on.  The synchronous
        trainer loses one step to it; the asynchronous one notices when its statistics window drains (at most STATS_WINDOW_PERSIST
        steps, each skipped on the device and rolled back individually).  Returns True if it fell back."""
        if ops.persist_starved(self.device):
            ops.degrade('one-launch BiLSTM layer', 'one launch per time step (ops.LSTM_PERSIST = False)')
            ops.LSTM_PERSIST = False
            return True
        return False

    def _skipped_on_device(self, st):
        """a step whose gradient norm turned out non-finite (read late, async_stats): the device skipped that update"""
        self.check_device_status()
        opt = getattr(self.optimizer, 'opt', None)
        if hasattr(opt, 'rollback_step'):
            # the host-side Adam step counts of the parameters of THAT update ran one ahead since then (bias corrections)
            opt.rollback_step(dict.get(st, 'opt_step'))
        self.verbose('Error : grad norm is not finite @ step %s (update skipped on the device)' % dict.get(st, 'step', '?'))

    def drain_stats(self, keep=0):
        """read the statistics of the steps issued so far -- all but the `keep` newest (waits for the GPU once: for the newest step that
        is read).  The device scalars of all those steps come back in ONE copy (one small gather launch) instead of one per scalar."""
        n = len(self._unread) - keep
        if n <= 0:
            return
        pending, self._unread = self._unread[:n], self._unread[n:]
        ringed = [st for st in pending if getattr(st, 'ring_slot', None) is not None]
        if ringed and self.__dict__.get('_stats_ring') is not None:
            # the ring comes to the host on a stream of its own, behind the event of the NEWEST step that is read: the host waits for that
            # step (finished a while ago when `keep` > 0), not for what has been issued since -- a read on the compute stream would be
            # ordered behind everything queued on it.  No device allocation either (see train_step).
            side = self._stats_stream
            side.wait_event(self._stats_events[ringed[-1].ring_slot])
            with torch.cuda.stream(side):
                self._stats_host.copy_(self._stats_ring, non_blocking=True)
            side.synchronize()
            host = self._stats_host.tolist()
            for st in ringed:
                r = host[st.ring_slot]
                for k, v in zip(st.ring_keys, r):
                    if torch.is_tensor(dict.__getitem__(st, k)):
                        st.put(k, v)
        for st in pending:
            st.materialise()
            self._stats_read(st)

    def _stats_read(self, st):
        """hook: the scalars of a step have just been read on the host"""

    def exec(self):
        t0 = time.perf_counter()
        frames = 0
        while self.step < self.max_step:
            text, sid, mel, linear = (t.to(self.device) for t in self.batches[self.step % len(self.batches)])
            st = self.train_step(text, sid, mel, linear)
            frames += mel.shape[0] * mel.shape[1]
            self.log.append(st)
            if self.step == 1 or self.step % 10 == 0:
                self.verbose('Tr stat | step %d | Loss - %.4f | Grad. Norm - %.3f | lr %.2e' %
                             (self.step, st['loss'], st['grad_norm'], st['lr']))
        torch.cuda.synchronize()
        self.drain_stats()
        self.check_device_status()
        dt = time.perf_counter() - t0
        if getattr(self.paras, 'save', False):
            self.save_checkpoint('latest.pth', self.log[-1]['loss'] if self.log else 0.0)
        self.verbose('%d steps, %d frames in %.2f s (%.0f frames/s incl. first-step set-up)' %
                     (len(self.log), frames, dt, frames / max(dt, 1e-9)))
        return self.log


EPS = 1e-10                      # ref: bin/train_vqvae.py:18


class VqvaeTrainer(TtsTrainer):
    """The speech -> text -> speech cycle of VqvaeTrainer.exec (bin/train_vqvae.py:139-176,208-270) on synthetic batches:

        pair_prob, _, unpair_prob, unpair_latent, unpair_len = model.speech_to_text(aug_mel, unpair_aug_mel)
        mel, linear, ..., unpair_mel, unpair_linear, ...     = model.text_to_speech(text, sid, unpair_sid, unpair_latent, ...)
        loss = asr_weight * CTC(log(pair_prob + EPS), text) + tts_weight * (freq_loss(mel) + freq_loss(linear))
               [+ unpair_speech_weight * (freq_loss(unpair_mel) + freq_loss(unpair_linear))]
        loss.backward(); all-reduce; clip_grad_norm_(5.0); optimizer.step()

    Everything between the inputs and the gradients runs on the HIP kernels: the CTC speech encoder and its backward
    (asr.py), the codebook lookup with the straight-through estimator (autograd.vq_l2), the run-length merge (autograd.mean_forward),
    the TTS branch, the CTC loss (autograd.ctc_loss)."""
    STATIC_GRAPH = False     # the cycles' autograd graphs depend on the data (ignore_speech_cycle, skip_prob draws, txt_update_codebook)

    def ctc_loss(self, prob, text):
        """compute_ctcloss with paras.actual_len = False (bin/train_vqvae.py:430-444): every frame counts, the targets are the
        non-zero tokens; torch.nn.CTCLoss() defaults (blank 0, mean over the batch of nll / target length)"""
        from . import autograd as AG
        return AG.ctc_loss(prob, text, EPS)

    def _async(self):
        from .optim import FusedAdam
        return self.async_stats and isinstance(getattr(self.optimizer, 'opt', None), FusedAdam)

    def _paired_losses(self, pair_prob, pair_post_prob, pm, pl, mel, linear, text, terms, stats, linear_loss=None):
        """the terms both cycles share (bin/train_vqvae.py:208-224): CTC on the paired posteriors (+ the ASRPostnet term) and
        freq_loss on the paired reconstruction.  Appends (weight, loss, statistics name) to `terms`: the weighted sum itself -- and the
        partial sums the log prints -- are ONE launch in _finish_step (the reference's chain of one-element kernels)."""
        hp = self.hp
        asr_loss = self.ctc_loss(pair_prob, text)                                                 # :209
        asr_w = float(hp.get('asr_weight', 1.0))
        if self.model.use_asr_postnet:                                                            # :210-213
            from . import autograd as AG
            pw = float(self.model.asr_postnet_weight)
            asr_post_loss = AG.ctc_loss(pair_post_prob, text, EPS, apply_log=False)               # compute_ctcloss(..., apply_log=False)
            terms += [(asr_w * (1.0 - pw), asr_loss, None), (asr_w * pw, asr_post_loss, None)]
            stats['asr_post_loss'] = asr_post_loss.detach()
        else:
            terms.append((asr_w, asr_loss, None))                                                 # :215
        # (:216-218: a NaN / inf CTC value is counted when the statistics are read -- as in the reference it is already inside total,
        # the gradient norm is then NaN and the update is skipped)
        stats['asr_loss'] = asr_loss.detach()
        # (linear_loss: the value _side_branch computed -- and sent backward -- on the postnet's stream)                           :221-224
        terms += [(self.tts_weight, self.freq_loss(pm, mel), 'tts_loss'),
                  (self.tts_weight, linear_loss if linear_loss is not None else self.freq_loss(pl, linear), 'tts_loss')]

    def _total(self, terms, stats):
        """total = sum of weight * loss over `terms`; the named partial sums (unweighted, as the reference logs them) land in `stats`"""
        from . import autograd as AG
        names = []
        for _, _, nm in terms:
            if nm is not None and nm not in names:
                names.append(nm)
        W = [[w for w, _, _ in terms]] + [[1.0 if nm == k else 0.0 for _, _, nm in terms] for k in names]
        outs = AG.scalar_combine(W, [x for _, x, _ in terms])
        for k, v in zip(names, outs[1:]):
            stats[k] = v
        return outs[0]

    def _count_ctc_nan(self, st):
        for k in ('asr_loss', 'unpair_text_loss'):
            if k in st and not math.isfinite(st[k]):
                self.ctc_nan = getattr(self, 'ctc_nan', 0) + 1

    def _side_branch(self, side_terms):
        """separate_postnet (src/tts.py:47-50) cuts the gradient between the decoder and the postnet: when Tacotron2 ran the postnet on the
        second stream (`postnet_stream`, TtsTrainer.set_model), the branch's losses (freq_loss of the linear spectrograms: `side_terms`) and its
        whole backward (CBHG incl. both GRU passes) are issued there NOW, beside whatever the main stream does next; _finish_step joins the
        streams after the main backward.  d total / d loss_i = w_i whatever the rest of the sum, so the two backward passes give the gradients
        of the one pass bit for bit (the branch's parameters get no other contribution).  Returns None when the postnet ran on the main stream,
        else (event, the loss values in the order of `side_terms`): the caller puts them where the one-stream step has these terms."""
        side = getattr(self.model.tts, 'postnet_stream', None)
        if side is None:
            return None
        with torch.cuda.stream(side):
            vals = [f() for _, f, _ in side_terms]
            total = self._total([(w, x, nm) for (w, _, nm), x in zip(side_terms, vals)], {})
            total.backward()
            ops.flush_wgrads()
            ev = self.__dict__.setdefault('_side_event', torch.cuda.Event())
            ev.record(side)
        ops.side_pending(ev)                      # (one-launch BiLSTM layers of the main stream wait for it: they need every compute unit)
        return ev, [x.detach() for x in vals]

    def _finish_step(self, terms, stats, tf_rate, kind, side=None):
        """BaseSolver.backward (src/solver.py:138-151) + the step counter.  Synchronous form: the scalars are read here, a NaN gradient
        norm skips the update (as the reference does).  async_stats: nothing is read -- the scalars go into the statistics ring, the
        guarded Adam skips a non-finite step on the device (TtsTrainer.train_step).  `side`: what _side_branch returned."""
        if side is None:
            total = self._total(terms, stats)
            total.backward()
        else:
            # the terms the second stream has already sent backward (no graph: detached values) stay out of this backward pass ...
            self._total([t for t in terms if t[1].requires_grad], {}).backward()
            torch.cuda.current_stream().wait_event(side[0])
            ops.side_pending(None)
            # ... and the reported sums are ONE launch over all terms in the order of the one-stream step: the same arithmetic, bit for bit
            with torch.no_grad():
                total = self._total([(w, x.detach(), nm) for w, x, nm in terms], stats)
        self._reduce_gradients()
        grad_norm = self._clip()
        lr = self.optimizer.lr_at(self.step)
        if self._async() and torch.is_tensor(grad_norm) and grad_norm.is_cuda:
            opt_step = getattr(self.optimizer.opt, 'guarded_steps', 0)
            self.optimizer.step(guard_norm=grad_norm)
            self.step += 1
            scalars = dict(stats, loss=total, grad_norm=grad_norm)
            return self._ring_stats(scalars, tf_rate=tf_rate, lr=lr, step=self.step - 1, opt_step=opt_step, kind=kind,
                                    **{k: v for k, v in self._step_info.items()})
        gn = float(grad_norm)
        if gn == gn:
            self.optimizer.step()
        else:
            self.check_device_status()                   # (a starved one-launch LSTM layer: per-step form from the next step on)
            self.verbose('Error : grad norm is NaN @ step ' + str(self.step))
        self.step += 1
        out = {k: float(v) for k, v in stats.items()}
        out.update(loss=float(total.detach()), grad_norm=gn, tf_rate=tf_rate, lr=lr, kind=kind, **self._step_info)
        self._count_ctc_nan(out)
        return out

    def _begin_step(self):
        from . import parallel
        parallel.collective_counts(reset=True)
        self._step_info = {}
        tf_rate = self.optimizer.pre_step(self.step)
        if getattr(self, 'reducer', None) is not None:
            self.reducer.prepare()
        return tf_rate

    def speech_first_step(self, mel, aug_mel, linear, text, sid, unpair_mel=None, unpair_aug_mel=None, unpair_linear=None,
                          unpair_sid=None, _masks=None, _asr_masks=None):
        """The speech -> text -> speech cycle of VqvaeTrainer.exec (bin/train_vqvae.py:159-176,208-233).  The only host read of the
        forward pass is the merged lengths of the unpaired latents (autograd.mean_forward): they fix the text length the TTS branch
        runs at (the reference reads every index, src/vqvae.py:225)."""
        hp = self.hp
        tf_rate = self._begin_step()
        pair_prob, _, unpair_prob, unpair_latent, unpair_latent_len, pair_post_prob, _ = self.model.speech_to_text(
            paired_mel=aug_mel, unpaired_mel=unpair_aug_mel, **({'_masks': _asr_masks} if _asr_masks is not None else {}))
        ignore_speech_cycle = unpair_latent is None                                               # :163-172
        if unpair_aug_mel is not None:
            self._step_info.update(unpair_text_len=0 if ignore_speech_cycle else int(unpair_latent.shape[1]))
        out = self._tts(text, sid, None if ignore_speech_cycle else unpair_sid, unpair_latent, None,
                                        unpair_latent_len, mel, None if ignore_speech_cycle else unpair_mel, tf_rate, _masks=_masks)
        pm, pl, _, _, upm, upl, _, _ = out
        # :232: the unpaired term only counts after the warm-up steps (weight 0 before: computed and logged, as in the reference)
        w = float(hp.get('unpair_speech_weight', 10.0)) if self.step > int(hp.get('unpair_speech_start_step', 0)) else 0.0
        side = self._side_branch([(self.tts_weight, lambda: self.freq_loss(pl, linear), 'tts_loss')] +
                                 ([] if ignore_speech_cycle else [(w, lambda: self.freq_loss(upl, unpair_linear), 'unpair_speech_loss')]))
        lin = side[1] if side is not None else [None, None]
        stats, terms = {}, []
        self._paired_losses(pair_prob, pair_post_prob, pm, pl, mel, linear, text, terms, stats, linear_loss=lin[0])
        if not ignore_speech_cycle:                                                               # :227-233
            terms += [(w, self.freq_loss(upm, unpair_mel), 'unpair_speech_loss'),
                      (w, lin[1] if side is not None else self.freq_loss(upl, unpair_linear), 'unpair_speech_loss')]
        return self._finish_step(terms, stats, tf_rate, 'speech_first', side)

    def text_first_step(self, mel, aug_mel, linear, text, sid, unpair_text=None, unpair_sid=None, _masks=None, _asr_masks=None):
        """The text -> speech -> text cycle of VqvaeTrainer.exec (bin/train_vqvae.py:186-205,208-224,234-250): text_to_speech on the
        paired text (teacher forced) and, when given, the unpaired text (rows without a teacher feed their own output back); the
        unpaired prediction is DETACHED and goes through speech_to_text next to the paired mel with `using_fake_mel` (the codebook
        table is detached for the fake part); losses: the paired terms, plus CTC of the unpaired posteriors against the unpaired text
        (paras.actual_len = False: every frame counts).  A NaN / inf unpaired term is counted and dropped, as the reference does."""
        hp = self.hp
        tf_rate = self._begin_step()
        use_unpair_text = unpair_text is not None                                                 # the caller gates it (:128,:149-152)
        asr = None
        if not use_unpair_text:
            # Without unpaired text the two halves of this cycle do not feed each other: the speech encoder goes FIRST (the reference runs it
            # second, :203-205), so that everything behind text_to_speech is backward work the postnet branch can run beside (_side_branch)
            asr = self.model.speech_to_text(paired_mel=aug_mel, unpaired_mel=None, using_fake_mel=False,
                                            **({'_masks': _asr_masks} if _asr_masks is not None else {}))
        out = self._tts(text, sid, unpair_sid if use_unpair_text else None, None, unpair_text, None, mel, None,
                                        tf_rate, _masks=_masks)                                   # :190-199
        pm, pl, _, _, upm, _, _, _ = out
        side = self._side_branch([(self.tts_weight, lambda: self.freq_loss(pl, linear), 'tts_loss')])
        if use_unpair_text:
            upm = upm.detach()                                                                    # :201-202
            asr = self.model.speech_to_text(paired_mel=aug_mel, unpaired_mel=upm, using_fake_mel=True,
                                            **({'_masks': _asr_masks} if _asr_masks is not None else {}))   # :203-205
        pair_prob, _, unpair_prob, _, _, pair_post_prob, _ = asr
        stats, terms = {}, []
        self._paired_losses(pair_prob, pair_post_prob, pm, pl, mel, linear, text, terms, stats, linear_loss=side[1][0] if side is not None else None)
        if use_unpair_text:                                                                       # :234-250
            ut = self.ctc_loss(unpair_prob, unpair_text)
            stats['unpair_text_loss'] = ut.detach()
            # :246-248: a non-finite unpaired term is dropped (and counted).  That decides what backward() sees, so it is the one scalar
            # this cycle reads inside the step (only configurations with unpair_text_weight > 0 get here; none of the shipped ones)
            v = float(ut.detach())
            if math.isfinite(v):
                terms.append((float(hp.get('unpair_text_weight', 0.0)), ut, None))
        return self._finish_step(terms, stats, tf_rate, 'text_first', side)

    def cycle_step(self, pair, unpair=None, _masks=None, _asr_masks=None):
        """One iteration of VqvaeTrainer.exec's loop body (bin/train_vqvae.py:124-150): even steps run the speech-first cycle, odd
        steps the text-first cycle; the unpaired batch joins only when its weight is positive and the step is past its start step.
        `pair` = (mel, aug_mel, linear, text, sid); `unpair` = the same five for the unpaired batch, or None."""
        mel, aug_mel, linear, text, sid = pair
        kw = dict(_masks=_masks, **({'_asr_masks': _asr_masks} if _asr_masks is not None else {}))
        kind, use_unpair = self.cycle_kind(self.step)
        if kind == 'speech_first':                                                                # :137
            if use_unpair and unpair is not None:
                umel, uaug, ulin, _, usid = unpair
                return self.speech_first_step(mel, aug_mel, linear, text, sid, unpair_mel=umel, unpair_aug_mel=uaug,
                                              unpair_linear=ulin, unpair_sid=usid, **kw)
            return self.speech_first_step(mel, aug_mel, linear, text, sid, **kw)
        if use_unpair and unpair is not None:
            _, _, _, utext, usid = unpair
            return self.text_first_step(mel, aug_mel, linear, text, sid, unpair_text=utext, unpair_sid=usid, **kw)
        return self.text_first_step(mel, aug_mel, linear, text, sid, **kw)

    def cycle_kind(self, step):
        """(which cycle step `step` runs, whether it takes an unpaired batch)        ref: bin/train_vqvae.py:128-129,137-150"""
        hp = self.hp
        if step % 2 == 0:
            return 'speech_first', float(hp.get('unpair_speech_weight', 10.0)) > 0 and step > int(hp.get('unpair_speech_start_step', 0))
        return 'text_first', float(hp.get('unpair_text_weight', 0.0)) > 0 and step > int(hp.get('unpair_text_start_step', 0))

    def _stats_read(self, st):
        self._count_ctc_nan(st)

    # -- the solver protocol of main.py:65-68 on synthetic batches
    def load_data(self):
        """Synthetic counterparts of the reference's pair_set / unpair_set loaders (bin/train_vqvae.py:55-69; src/data.py is
        outside the path): `n_batches` seeded batches each, a different seed per rank (utterance-level data parallelism), mel / linear
        zero-padded to a multiple of r and aug_mel unpadded exactly as fetch_data delivers them (:33-53).  The unpaired set has its
        own batch size and length (--unpair-batch-size / --unpair-frames; default: the paired ones)."""
        from .synthetic import synthetic_cycle_batch
        pa = self.paras
        B = int(getattr(pa, 'batch_size', None) or self.config['data']['corpus'].get('batch_size', 8))
        Bu = int(getattr(pa, 'unpair_batch_size', None) or B)
        frames = int(getattr(pa, 'frames', 256))
        uframes = int(getattr(pa, 'unpair_frames', None) or frames)
        n = int(getattr(pa, 'n_batches', 1))
        self.r = self.config['model']['decoder']['decoder']['n_frames_per_step']
        rank = int(os.environ.get('RANK', 0))
        seed = getattr(pa, 'seed', 0)
        lo, hi = self.config['data']['audio'].get('time_stretch_range', [1.0, 1.0]) if getattr(pa, 'stretch', False) else (1.0, 1.0)
        rs = np.random.RandomState(seed + 77 + rank)
        mk = lambda b, f, sd: synthetic_cycle_batch(b, f, self.r, self.vocab_size, self.n_spkr, self.n_mels, self.linear_dim, seed=sd,
                                                    stretch=float(rs.uniform(lo, hi)))
        self.pair_set = [mk(B, frames, 1000 * rank + i + seed) for i in range(n)]
        self.unpair_set = [mk(Bu, uframes, 500000 + 1000 * rank + i + seed) for i in range(n)]
        self.pair_iter, self.unpair_iter = 0, 0
        # (mel, aug_mel, linear, text, sid) -> the paired TTS step's (text, sid, mel, linear): TtsTrainer.exec on the same data
        self.batches = [(b[3], b[4], b[0], b[2]) for b in self.pair_set]
        return self

    def fetch_data(self, iter_name):
        """the next batch of `pair_iter` / `unpair_iter` on the device, the set restarting when it is exhausted (:33-41)"""
        data = getattr(self, iter_name.replace('iter', 'set'))
        i = getattr(self, iter_name)
        setattr(self, iter_name, i + 1)
        cache = self.__dict__.setdefault('_dev_batches', {})
        key = (iter_name, i % len(data))
        if key not in cache:                     # (the synthetic sets are small and fixed: on the device once)
            cache[key] = tuple(t.to(self.device) for t in data[i % len(data)])
        return cache[key]

    def exec(self):
        """VqvaeTrainer.exec's loop (bin/train_vqvae.py:111-150,270-300) without the corpus-side logging: paired batch every step,
        the unpaired batch fetched only when the step's cycle uses it, cycles alternating"""
        t0 = time.perf_counter()
        frames = 0
        cnt = {'unp_sph': 0, 'unp_txt': 0}
        self.ctc_nan = 0
        while self.step < self.max_step:
            pair = self.fetch_data('pair_iter')
            kind, use_unpair = self.cycle_kind(self.step)
            unpair = self.fetch_data('unpair_iter') if use_unpair else None                       # :139-150
            st = self.cycle_step(pair, unpair)
            frames += pair[0].shape[0] * pair[0].shape[1] + (unpair[0].shape[0] * unpair[0].shape[1] if unpair is not None else 0)
            if unpair is not None:
                cnt['unp_sph' if kind == 'speech_first' else 'unp_txt'] += 1
            self.log.append(st)
            if self.step == 1 or self.step % 10 == 0:
                self.verbose('Tr stat | step %d (%s) | Loss - %.4f (CTC-nan/unp-sph/unp-txt=%d/%d/%d) | Grad. Norm - %.3f | lr %.2e' %
                             (self.step, kind, st['loss'], self.ctc_nan, cnt['unp_sph'], cnt['unp_txt'], st['grad_norm'], st['lr']))
        torch.cuda.synchronize()
        self.drain_stats()
        self.check_device_status()
        dt = time.perf_counter() - t0
        if getattr(self.paras, 'save', False):
            self.save_checkpoint('latest.pth', self.log[-1]['loss'] if self.log else 0.0)
        self.verbose('%d steps, %d frames in %.2f s (%.0f frames/s incl. first-step set-up)' %
                     (len(self.log), frames, dt, frames / max(dt, 1e-9)))
        return self.log
