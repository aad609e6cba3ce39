#!/usr/bin/env python3
"""Headline benchmark: mel-frames/sec of the decode+attention region (Decoder.forward:
processed memory + AdaIN statistics + every autoregressive decode step with location-sensitive
attention), free-running inference, on BASELINE.json config 2:
B = 32 utterances per GPU, 256 -> 258 frames (86 decode steps, r = 3), L = 43, n_mels = 80, fp32,
prenet dropout 0.5 active (it never turns off in the reference), synthetic weights and inputs.

    python bench.py [--gpus N --steps K --warmup W] [--workload c2|c5|c3|train]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch.  Weak scaling: every rank works on its own batch
(utterances are independent; inference has no data-path collective, training all-reduces gradients).
With --gpus N > 1 and no WORLD_SIZE in the environment this process starts N ranks itself
(python -m torch.distributed.run, one per GPU, RCCL) BEFORE touching the GPU, and exits with their code.
Prints ONE JSON line on rank 0.

Secondary workloads (not the headline):
  c5     BASELINE config 5 (bin/gen_specgram.py long form: B=64, L=171, 355 decode steps)
  c3     the VQ nearest-code search (L2Embedding.forward) at 32x129 / 256x129 vectors, V=43 / 512
  train  BASELINE config 4: the paired TTS training step, B=32 per GPU, 109 speakers, SyncBN over the
         global batch, RCCL all-reduce of the gradients (ms_allreduce / ms_syncbn reported)
  cycle  BASELINE config 3 as the reference trains it: VqvaeTrainer.exec's alternating speech-first / text-first
         cycles (CTC speech encoder + VQ search + run-length merge + TTS branch), paired + unpaired batch
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

B, T_RAW, R, L, N_MELS = 32, 256, 3, 43, 80
T = T_RAW + (R - T_RAW % R)          # 258: the reference pads r - T % r frames (bin/train_vqvae.py:43-46)
STEPS = T // R                        # 86
MFMA_F32_PEAK_TFLOPS = 157.3          # MI355X fp32 matrix peak (256 CUs x 256 FLOP/cycle x 2.4 GHz)
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
# PMC pass of the dominant kernel (tools/gpu_pmc.sh -> tools/pmc_summary.py): HBM bytes per launch.  bench.py does not
# measure this itself (PMC counters need their own rocprofv3 passes); the value is quoted WITH its source file.
TRAFFIC_FILES = ('profiles/r06_pmc_hbm_traffic%s.json', 'profiles/r05_pmc_hbm_traffic%s.json', 'profiles/r04_pmc_hbm_traffic%s.json', 'profiles/r03_pmc_hbm_traffic%s.json', 'profiles/r02_pmc_hbm_traffic%s.json', 'profiles/r01_pmc_hbm_traffic%s.json')


def decode_step_algorithmic_bytes(Bsz, Lt, dec):
    """bytes one decode step must move if every operand is read once: all per-step weights, the memory and
    processed-memory tiles of every utterance, and the recurrent state (read + write)."""
    P, Q, D, E, A = dec.prenet_dim, dec.query_rnn_dim, dec.dec_rnn_dim, dec.enc_embed_dim, dec.attn_dim
    in_dim = dec.n_mels * dec.n_frames_per_step
    F_, K = dec.n_location_filters, dec.location_kernel_size
    w = (in_dim * P + P * P                       # prenet
         + 4 * Q * (P + E + Q) + 8 * Q            # query LSTM
         + A * Q + F_ * 2 * K + A * F_ + A        # attention: query layer, location conv, location linear, v
         + 4 * D * (E + Q + D) + 8 * D            # decoder LSTM
         + (in_dim + 1) * (D + E) + in_dim + 1    # proj + gate
         + 2 * (dec.spkr_embed_dim * Q + Q))      # AdaIN linears: counted per step as SURVEY 8d does (18,876,337 weights =
    #                                               75.5 MB), although this implementation hoists them out of the loop
    mem = Bsz * Lt * (E + A)
    state = Bsz * (2 * (2 * Q + 2 * D) + 2 * E + 4 * Lt + 2 * in_dim)
    return 4.0 * (w + mem + state)


def lstm_algorithmic_bytes(Bsz, H, K):
    """bytes one LSTM-cell launch must move: weights (4H x K) + biases + x/h in + c in/out + h out"""
    return 4 * (4 * H * K + 8 * H + Bsz * K + 3 * Bsz * H)


def _quoted_traffic(sfx):
    """roofline.traffic from the committed PMC summary of this workload (tools/gpu_pmc.sh + tools/pmc_summary.py), with its source"""
    for f in TRAFFIC_FILES:
        rel = f % sfx
        try:
            with open(os.path.join(REPO, rel)) as fh:
                tj = json.load(fh)
            return {'traffic': tj['hbm_bytes_per_launch'],
                    'traffic_source': {'file': rel, 'kernel': tj.get('kernel'), 'commit': tj.get('commit'),
                                       'note': 'separate rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE), not measured by this run'}}
        except (OSError, KeyError, ValueError):
            continue
    return {'traffic': None}


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_timed(short_fn, full_fn, units_per_pass, unit, kind, sample, max_passes=3, budget=(25.0, 45.0)):
    """CPU baseline protocol shared by every workload: B=32 GEMMs stop scaling long before a 128-core host is full (and
    oversubscribed thread pools get pathologically slow), so `short_fn()` (a few per cent of a pass) is timed at 8 ... 64 threads
    first; `full_fn(i)` (one full pass of the workload) then runs 1 warm-up + up to `max_passes` timed passes at the two fastest
    settings, and the better median is the baseline.  Bounded: the probe stops after 20 s, the passes after `budget` seconds."""
    import numpy as np
    import torch
    cores = host_cores()
    probe = []
    t_start = time.perf_counter()
    with torch.no_grad() if kind != 'nn_modules_autograd' else torch.enable_grad():
        for threads in sorted({t for t in (8, 16, 32, 64, min(cores, 64)) if t <= cores}):   # > 64 threads never won (r04: 13.7 s at 256)
            torch.set_num_threads(threads)
            short_fn()                                             # warm-up of the thread pool at this size
            t0 = time.perf_counter()
            short_fn()
            probe.append({'threads': threads, 'probe_ms': round((time.perf_counter() - t0) * 1e3, 3)})
            if time.perf_counter() - t_start > 20.0:
                break
        cand = [r['threads'] for r in sorted(probe, key=lambda r: r['probe_ms'])[:2]]
        full = {}
        for th in cand:
            torch.set_num_threads(th)
            times = []
            for i in range(1 + max_passes):
                t0 = time.perf_counter()
                full_fn(i)
                times.append(time.perf_counter() - t0)
                if time.perf_counter() - t_start > (budget[0] if th == cand[0] else budget[1]) and len(times) >= 2:
                    break
            full[th] = times
        best = min(full, key=lambda th: float(np.median(full[th][1:])))
        times = full[best]
    t = float(np.median(times[1:]))
    return {'value': units_per_pass / t, 'unit': unit, 'cores': best, 'kind': 'port', 'assembly': kind.replace('_autograd', ''), 'host_cores': cores,
            'seconds_per_pass': t, 'thread_probe': probe,
            'full_pass_seconds': {str(k): [round(x, 4) for x in v] for k, v in full.items()},
            'sample': sample + '; %d timed pass(es) after 1 warm-up, median; torch CPU fp32, %d threads = the better of the two fastest '
                               'settings of the probe in `thread_probe` (host has %d cores)' % (len(times) - 1, best, cores)}


def cpu_baseline(m, txt_mem, spk, frames=None):
    """The CPU reference of the decode workload on this box's host cores: the decode path assembled from the torch.nn modules
    the reference itself is made of (nn.LSTMCell, nn.Linear, nn.Conv1d: oracle/nn_baseline.py, the same ATen kernels the
    reference would hit)."""
    from oracle import nn_baseline as NB
    from helpers import full_hp
    W = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    mem, s = txt_mem.cpu(), spk.cpu()
    ref = NB.NNDecoder(W, full_hp(0.5))
    frames = frames or T
    Bn, Ln = mem.shape[0], mem.shape[1]
    return cpu_timed(lambda: ref(mem, 3 * R, s), lambda i: ref(mem, frames, s, seed=i), Bn * frames, 'mel-frames/s', 'nn_modules',
                     'full passes of Decoder.forward (B=%d, %d steps, L=%d, prenet dropout 0.5); the decode loop assembled from the '
                     'torch.nn modules the reference is made of (nn.LSTMCell / nn.Linear / nn.Conv1d, oracle/nn_baseline.py); probe = 3 '
                     'decode steps' % (Bn, frames // R, Ln), max_passes=3 if Bn * frames <= 32 * 300 else 2)


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _has_result_line(text):
    return any(l.startswith('{') and '"metric"' in l for l in text.splitlines())


def spawn_ranks(args):
    """--gpus N > 1 without a launcher: start N fresh ranks (one per GPU) before this process touches the GPU.  Should the
    rank set die without printing its JSON line (a launcher or rendezvous problem the ranks' own RCCL probe cannot see), a
    SECOND set of fresh ranks runs on gloo: the replica workloads only need a barrier and a max, the training workload
    stages its all-reduce through the host (slow, flagged in the line).  No rank is ever re-executed."""
    import torch
    n_dev = torch.cuda.device_count()                  # counting devices does not initialise the GPU
    if args.gpus > n_dev and os.environ.get('ST_BENCH_BACKEND', 'nccl') == 'nccl':
        sys.stderr.write('bench.py: --gpus %d but this node exposes %d GPU(s): refusing to run fewer ranks than asked '
                         '(set ST_BENCH_BACKEND=gloo to share devices in a functional test)\n' % (args.gpus, n_dev))
        return 2
    env = dict(os.environ)
    # This pool's host driver only supports dmabuf IPC: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL's peer-memory setup fails
    # with `hipIpcGetMemHandle: invalid argument` (environment notes of the build image, which exports it already; kept as a
    # default so that a launcher with a scrubbed environment still works).
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

    def run_set(e):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        r = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, text=True)
        sys.stdout.write(r.stdout)
        sys.stdout.flush()
        return r.returncode, r.stdout
    rc, out = run_set(env)
    if (rc != 0 or not _has_result_line(out)) and env.get('ST_BENCH_BACKEND', 'nccl') == 'nccl':
        sys.stderr.write('bench.py: the rank set exited with %d and no result line; second set of fresh ranks on gloo\n' % rc)
        env2 = dict(env, ST_BENCH_BACKEND='gloo', ST_BENCH_RCCL_NOTE='first rank set exited with code %d before printing a line' % rc)
        rc, out = run_set(env2)
    return rc


def rccl_probe_main():
    """`bench.py --rccl-probe` (a CHILD of every rank, started before the rank touches the GPU): initialise RCCL with the rank's
    coordinates on a port of its own, run one all-reduce over a device tensor and check the sum.  Exit code 0 = RCCL works on
    this node; anything else (exception, crash, the parent's time-out on a hang) = it does not, and the ranks go on with gloo."""
    fake = os.environ.get('ST_BENCH_FAKE_RCCL_FAIL')            # tests: what a failing peer-memory set-up looks like from outside
    if fake == '1' or fake == 'rank:' + os.environ.get('RANK', ''):      # ('rank:3': on that rank only)
        sys.stderr.write('rccl probe: failure faked by ST_BENCH_FAKE_RCCL_FAIL\n')
        sys.exit(3)
    if os.environ.get('ST_BENCH_FAKE_RCCL_OK') == '1':          # tests on a box without GPUs: every other rank's probe "passes"
        print('RCCL_PROBE_OK')
        return
    import datetime
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ.get('LOCAL_RANK', '0'))
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        sys.stderr.write('rccl probe: no GPU visible\n')
        sys.exit(4)
    dev = torch.device('cuda', local % n_dev)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=90))
    t = torch.full((1 << 16,), float(rank + 1), device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    want = world * (world + 1) / 2.0
    if float(t[0].item()) != want or float(t[-1].item()) != want:
        sys.stderr.write('rccl probe: all-reduce returned %r, expected %r\n' % (float(t[0].item()), want))
        sys.exit(5)
    dist.barrier()
    dist.destroy_process_group()
    print('RCCL_PROBE_OK')


class Ranks:
    """rank bookkeeping + the barrier / max-over-ranks timing contract.

    N > 1: the ranks first meet on a TCP store (no GPU, no RCCL).  Every rank then runs `bench.py --rccl-probe` as a CHILD process
    (this process has not touched the GPU yet) and posts the outcome; only when EVERY rank's probe passed is the process group
    created on RCCL ('nccl' on ROCm).  Otherwise -- peer-memory set-up failing, a hang (the child is killed after
    ST_BENCH_PROBE_TIMEOUT seconds), a crash -- the group is created on gloo: the replica workloads' only collectives are the
    barrier and the max of the timing contract, so their line is produced either way, with `rccl_ranks: 0` and the probe's error
    in `collectives`; the training workload then stages its reductions through the host and says so."""

    def __init__(self, args):
        import torch
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        if self.world != args.gpus:
            raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, self.world))
        self.cpu_only = os.environ.get('ST_BENCH_ALLOW_CPU') == '1' and torch.cuda.device_count() == 0      # tests of this class
        # ST_BENCH_BACKEND=gloo: skip the probe (a box with fewer GPUs than ranks: the ranks share devices, reductions go through
        # CPU tensors; also the second rank set of spawn_ranks)
        self.backend = os.environ.get('ST_BENCH_BACKEND', 'nccl')
        self.probe = os.environ.get('ST_BENCH_RCCL_NOTE') or ('not run (ST_BENCH_BACKEND=gloo)' if self.backend == 'gloo' else None)
        n_dev = torch.cuda.device_count()
        if self.backend == 'nccl' and self.world > max(n_dev, 1) and not self.cpu_only:
            raise SystemExit('bench.py: %d ranks but %d GPU(s)' % (self.world, n_dev))
        self.dist = None
        # --dist (or ST_BENCH_FORCE_DIST=1): initialise the process group even for ONE rank and force every collective to be
        # issued (parallel.force_collectives), so a 1-GPU box executes the RCCL branch end to end -- world-size-1 sums are the
        # identity, the numbers must equal the plain run's.  This process has not touched the GPU yet.
        self.forced = self.world == 1 and (getattr(args, 'dist', False) or os.environ.get('ST_BENCH_FORCE_DIST') == '1')
        store = None
        if self.world > 1 or self.forced:
            import datetime
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if self.forced:
                os.environ.setdefault('MASTER_PORT', str(free_port()))
            # env:// rendezvous: under torch.distributed.run the launcher's agent hosts the store and every rank is a client
            # (TORCHELASTIC_USE_AGENT_STORE), otherwise rank 0 hosts it
            store, _, _ = next(dist.rendezvous('env://', rank=self.rank, world_size=self.world))
            store.set_timeout(datetime.timedelta(seconds=900))
            if self.backend == 'nccl' and self.world > 1:
                self.backend, self.probe = self._probe_rccl(store)
        if not self.cpu_only:
            assert torch.cuda.is_available(), 'bench.py needs a GPU'
            self.local_dev = self.local_rank % n_dev
            torch.cuda.set_device(self.local_dev)
            self.dev = torch.device('cuda', self.local_dev)
        else:
            self.dev = torch.device('cpu')
        if store is not None:
            kw = {'device_id': self.dev} if self.backend == 'nccl' else {}
            dist.init_process_group(self.backend, store=dist.PrefixStore('pg_' + self.backend, store), rank=self.rank,
                                    world_size=self.world, **kw)
            self.dist = dist
            if self.forced:
                from semi_tts_amd import parallel
                parallel.force_collectives(True)

    def _probe_rccl(self, store):
        """-> (backend, note): 'nccl' when every rank's child proved RCCL, else 'gloo' + the first error"""
        if self.rank == 0:
            store.set('probe_port', str(free_port()))
        port = store.get('probe_port').decode()
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=port, RANK=str(self.rank), WORLD_SIZE=str(self.world),
                   LOCAL_RANK=str(self.local_rank))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        limit = float(os.environ.get('ST_BENCH_PROBE_TIMEOUT', '300'))
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--rccl-probe'], env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, text=True, timeout=limit)
            ok = r.returncode == 0 and 'RCCL_PROBE_OK' in r.stdout
            msg = 'ok' if ok else 'rank %d: exit %d: %s' % (self.rank, r.returncode, (r.stderr.strip().splitlines() or ['?'])[-1][:300])
        except subprocess.TimeoutExpired:
            ok, msg = False, 'rank %d: no answer within %.0f s (killed)' % (self.rank, limit)
        store.set('probe_%d' % self.rank, msg)
        msgs = [store.get('probe_%d' % r).decode() for r in range(self.world)]      # (blocks until every rank has posted)
        bad = [m for m in msgs if m != 'ok']
        if bad:
            if self.rank == 0:
                sys.stderr.write('bench.py: RCCL probe failed (%s); continuing on gloo\n' % bad[0])
            return 'gloo', 'failed: ' + bad[0]
        return 'nccl', 'ok'

    def collectives_flat(self):
        """what the JSON line says about the process group: `rccl_ranks` (0 = no RCCL group) + `collectives`"""
        n = self.world if (self.dist is not None and self.backend == 'nccl') else 0
        return {'rccl_ranks': n, 'collectives': {'backend': self.backend if self.dist is not None else None,
                                                 'rccl_probe': self.probe if self.world > 1 else None}}

    def sync(self):
        import torch
        if not self.cpu_only:
            torch.cuda.synchronize()

    def barrier(self):
        self.sync()
        if self.dist is not None:
            self.dist.barrier()
        self.sync()

    def max_seconds(self, seconds):
        import torch
        if self.dist is None:
            return seconds
        t = torch.tensor([seconds], device=self.dev if self.backend == 'nccl' else 'cpu', dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, fn, steps, warmup):
        """W untimed + exactly K timed calls of fn, bracketed by barrier + synchronize; max over ranks"""
        for _ in range(warmup):
            fn()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        self.issue_seconds = time.perf_counter() - t0      # (diagnostic: how long the host needed to ISSUE the K steps; ~ the total = host-bound)
        self.barrier()
        return self.max_seconds(time.perf_counter() - t0)

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def event_timer(lib):
    """HIP events on the stream the st_* kernels are launched on (torch.cuda.Event would also do for torch's current stream;
    this goes through the library's own plumbing so it is right under ops.use_stream as well)"""
    import ctypes as C
    from semi_tts_amd import ops

    class Timer:
        def __enter__(self):
            self.e0, self.e1 = C.c_void_p(), C.c_void_p()
            lib.st_event_create(C.byref(self.e0)); lib.st_event_create(C.byref(self.e1))
            lib.st_event_record(self.e0, ops.stream_handle())
            return self

        def __exit__(self, *a):
            lib.st_event_record(self.e1, ops.stream_handle())
            ms = C.c_float()
            lib.st_event_elapsed_ms(self.e0, self.e1, C.byref(ms))     # synchronises on e1
            self.ms = ms.value
            lib.st_event_destroy(self.e0); lib.st_event_destroy(self.e1)
    return Timer


# ===================================================================================== decode (c2 / c5)
def lstm_probe(dec, dev, Bsz):
    """Roofline probe of the dominant kernel (the weight-streaming LSTM cell), measured with HIP events on the stream the
    kernel runs on: the two launches of a decode step alternate exactly as in the loop -- query LSTM K=1792; decoder LSTM
    K=2560, or (round 6, 16 < B <= 32: Decoder.split_gates) the first half of its reduction (Decoder.gate_split_k: 1280 columns) + the slab
    of the other half that rode beside the pq / fin launch -- so the weights cycle through the caches as they do there.  Replayed from a hipGraph (as in
    the decode loop) so the measurement sees device time, not the Python/ctypes issue rate."""
    import torch
    from semi_tts_amd import _lib, ops
    lib = _lib.load()
    Q, D, E, P = dec.query_rnn_dim, dec.dec_rnn_dim, dec.enc_embed_dim, dec.prenet_dim
    f32 = dict(device=dev, dtype=torch.float32)
    wq_ih, wd_ih = dec.query_rnn.weight_ih, dec.dec_rnn.weight_ih
    pk_q = ops.pack_weight([wq_ih, wq_ih[:, P:], dec.query_rnn.weight_hh], [P, E, Q], 4 * Q, lstm_H=Q, ldws=[P + E, P + E, Q])
    pk_d = ops.pack_weight([wd_ih, wd_ih[:, E:], dec.dec_rnn.weight_hh], [E, Q, D], 4 * D, lstm_H=D, ldws=[E + Q, E + Q, D])
    Kq, Kd = P + E + Q, E + Q + D                       # all multiples of 16 at the headline shape
    xq = ops.tile_rows(torch.randn(Bsz, Kq, **f32))
    xd = ops.tile_rows(torch.randn(Bsz, Kd, **f32))
    c_q, c_d = torch.randn(Bsz, Q, **f32), torch.randn(Bsz, D, **f32)
    ho, co = torch.zeros(ops.t16_floats(Bsz, Q), **f32), torch.empty(Bsz, Q, **f32)
    xq_v, xd_v, ho_v = ops.t16_view(xq, K=Kq), ops.t16_view(xd, K=Kd), ops.t16_view(ho, K=Q)

    split = bool(getattr(dec, 'split_gates', False)) and 16 < Bsz <= 32 and Q % 16 == 0 and D % 16 == 0 and E % 16 == 0
    slab = torch.randn(Bsz, 4 * D, **f32) if split else None
    Kc = dec.gate_split_k() if split else Kd

    def pair():
        ops.lstm_cell_packed(pk_q, xq_v, Kq, dec.query_rnn.bias_ih, dec.query_rnn.bias_hh, c_q, ho_v, co, Bsz, Q)
        if split:
            ops.lstm_cell_packed_part(pk_d, Kd // 16, xd_v, Kc, slab, dec.dec_rnn.bias_ih, dec.dec_rnn.bias_hh, c_d, ho_v, co, Bsz, D)
        else:
            ops.lstm_cell_packed(pk_d, xd_v, Kd, dec.dec_rnn.bias_ih, dec.dec_rnn.bias_hh, c_d, ho_v, co, Bsz, D)

    inner, outer = 50, 10

    def timed_graph(body):
        g2 = ops.Graph()
        body()
        with g2.capture():
            for _ in range(inner):
                body()
        for _ in range(3):
            g2.launch()
        torch.cuda.synchronize()
        with event_timer(lib)() as tm:
            for _ in range(outer):
                g2.launch()
        return tm.ms * 1e3 / (inner * outer)

    # (washing the caches between the cells, or having their operands written by a launch just before them, does not change the probe:
    # 5.75-5.79 us either way; in situ the two launches take 7.4 + 5.2 us (kernel trace medians): the first loads of a launch that follows
    # another one's stores cost more than back-to-back launches of the same kernel show)
    avg_us = timed_graph(pair) / 2
    Kd_cell = Kc                                        # (what the decoder CELL launch reduces; + the slab it reads)
    alg = 0.5 * (lstm_algorithmic_bytes(Bsz, Q, Kq) + lstm_algorithmic_bytes(Bsz, D, Kd_cell) + (4.0 * Bsz * 4 * D if split else 0.0))
    flops = 0.5 * (2.0 * Bsz * 4 * Q * Kq + 2.0 * Bsz * 4 * D * Kd_cell)
    return avg_us, alg, flops


def kernel_name_of_lstm(Bsz):
    bt = (Bsz + 15) // 16
    if bt in (2, 4):
        return 'pk_lstm_rt2_kernel<8,2,%d> (fused LSTM cell on packed operands: gate GEMM + pointwise; 2 row tiles x %d batch tile(s) per workgroup)' % (bt // 2, bt // 2)
    return 'pk_kernel<0,%d,8,2> (fused LSTM cell on packed operands)' % bt


def bench_decode(args, rk):
    import torch
    global B, L, T, STEPS
    if args.workload == 'c5':
        B, L, STEPS = 64, 171, (1026 + 40) // R
        T = STEPS * R
    dev = rk.dev
    from helpers import full_tacotron
    from semi_tts_amd import ops
    from semi_tts_amd.runtime import GraphedDecoder
    from semi_tts_amd.synthetic import synthetic_batch

    m = full_tacotron(dev, seed=1234, prenet_dropout=0.5)
    txt, spk, _ = synthetic_batch(B, L, T, seed=100 + rk.rank)
    txt, spk = torch.from_numpy(txt).to(dev), torch.from_numpy(spk).to(dev)
    with torch.no_grad():
        memory = m.encoder(txt, None).contiguous()        # inputs of the timed region, resident in HBM
    dec = m.decoder
    if args.pre_parts:
        dec.attn_pre_parts = args.pre_parts
    gd = GraphedDecoder(dec, B, L, T, dev)
    gd.memory.copy_(memory)
    gd.spkr.copy_(spk)
    graph = None
    if not args.no_graph:
        gd.capture()
        graph = gd.graph
    state = {}

    def one_pass():
        # fresh prenet dropout masks every pass (device RNG, inside the timed region), then the loop
        if graph is not None:
            state['out'] = gd(redraw=True)
        else:
            gd.draw_masks()
            state['out'] = gd._run()

    one_pass()
    elapsed = rk.timed(one_pass, args.steps, args.warmup)
    mel = state['out'][0]
    assert args.no_finite_check or bool(torch.isfinite(mel).all()), 'non-finite mel output'
    if not args.no_finite_check:
        # a starved in-launch hand-off makes a replay NaN (GraphedDecoder.check() would recover, but a timed region with such a replay
        # in it is not a measurement)
        assert not ops.handoff_starved(dec.handoff_status), 'a timed replay was starved of compute units (shared GPU?)' 
    if rk.rank != 0:
        return None

    us_step = elapsed / args.steps / STEPS * 1e6
    avg_us, alg, flops = lstm_probe(dec, dev, B)
    achieved = alg / (avg_us * 1e-6) / 1e9
    traffic, traffic_src = None, None
    if True:
        sfx = '' if args.workload == 'c2' else '_' + args.workload
        for rel in ([args.traffic_json] if args.traffic_json else []) + [f % sfx for f in TRAFFIC_FILES]:
            try:
                with open(rel if os.path.isabs(rel) else os.path.join(REPO, rel)) as f:
                    tj = json.load(f)
                traffic = tj['hbm_bytes_per_launch']
                traffic_src = {'file': rel, 'kernel': tj.get('kernel'), 'commit': tj.get('commit'),
                               'note': 'separate rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE, tools/gpu_pmc.sh), not measured by this run'}
                break
            except (OSError, KeyError, ValueError):
                continue
    step_bytes = decode_step_algorithmic_bytes(B, L, dec)
    split_on = bool(getattr(dec, 'split_gates', False)) and 16 < B <= 32
    kc = dec.gate_split_k() if split_on else 0
    kd_all = dec.enc_embed_dim + dec.query_rnn_dim + dec.dec_rnn_dim
    roof = {'bound': 'hbm', 'kernel': kernel_name_of_lstm(B) + (('; two launches per step: the query cell (K = 1792) and the decoder cell over the first %d columns of its '
                                                                  'reduction + the slab of the other %d that ride beside the pq / fin launch (pk_attnfin_part_kernel, '
                                                                  '%.1f MB of weights on 128 compute units)') % (kc, kd_all - kc, 16.0 * dec.dec_rnn_dim * (kd_all - kc) / 1e6)
                                                                 if split_on else ''),
            'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': traffic_src,
            'algorithmic_bytes_per_launch': alg, 'avg_launch_us': round(avg_us, 3),
            'launches_per_step': 2 * STEPS,
            # the same launch against the fp32 matrix-core peak (at B=32 the cell sits just left of the ridge:
            # 15.7 FLOP/B against 19.7)
            'mfma': {'achieved': round(flops / (avg_us * 1e-6) / 1e12, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(flops / (avg_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)},
            # the WHOLE decode step (all launches of one step; they are dependent, so the step is their sum) against the
            # same HBM peak: this, not the dominant kernel's fraction, is how far the path is from its roofline
            'step': {'algorithmic_bytes': step_bytes, 'us': round(us_step, 2),
                     'achieved': round(step_bytes / (us_step * 1e-6) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(step_bytes / (us_step * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}}
    frames = rk.world * B * T * args.steps
    res = {
        'metric': 'mel-frames/sec (decode+attn)', 'value': round(frames / elapsed, 1), 'unit': 'mel-frames/s',
        'n_gpus': rk.world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s: Decoder.forward free-running inference (decode+attn), B=%d per GPU, '
                               '%s->%d frames = %d decode steps (r=3), L=%d, n_mels=%d, prenet dropout 0.5, '
                               'config/supervised.yaml decoder section'
                               % (args.workload.upper(), B, '256' if args.workload == 'c2' else '1026+40', T, STEPS, L, N_MELS),
                   'batch_per_gpu': B, 'frames': T, 'decode_steps': STEPS, 'text_len': L,
                   'parallelism': 'replicas x%d (utterance-sharded, no collective)' % rk.world,
                   'launch': 'eager' if graph is None else 'hipGraph replay of the whole decode loop'},
        **rk.collectives_flat(),
        'us_per_decode_step': round(us_step, 2),
        'roofline': roof,
    }
    if not args.no_cpu_baseline and rk.world == 1:      # (rank 0 at N = 1 only: the other ranks of a multi-GPU run would idle in the barrier)
        res['cpu_baseline'] = cpu_baseline(m, memory, spk, frames=T)
    try:
        res['device'] = ops.device_info()
    except Exception:
        pass
    return res


# ===================================================================================== VQ (c3)
def bench_vq(args, rk):
    """L2Embedding.forward's nearest-code search (src/embed.py:105-147,208-213) on (B,129,64) encoder latents:
    BASELINE config 3 (V=512 synthetic table) and the native V=43 table, at 32 and 256 utterances.
    Algorithmic bytes per vector: 256 in + 256 out + 8 idx + 4V p_code; the table is read once."""
    import torch
    from semi_tts_amd import _lib, ops
    lib = _lib.load()
    dev = rk.dev
    g = torch.Generator().manual_seed(7 + rk.rank)
    rows = []
    head = None
    for Bn, V in (((32, 512),) if args.vq_head_only else ((32, 512), (256, 512), (32, 43), (256, 43))):
        n, D = Bn * 129, 64
        x = torch.randn(Bn, 129, D, generator=g).to(dev)
        table = torch.randn(V, D, generator=g).to(dev)
        temp = torch.ones(1, device=dev)
        p, idx, out = ops.vq_l2(x, table, temp)
        p_buf, idx_buf, out_buf = torch.empty_like(p), torch.empty_like(idx), torch.empty_like(out)
        ws = torch.empty(int(lib.st_vq_l2_workspace_floats(D, V)), device=dev)
        packed = ops.vq_pack_table(table)          # once per table version (embed.L2Embedding caches it): NOT part of a lookup

        def launch():                              # what L2Embedding.forward issues per call in frozen-weight inference
            lib.st_vq_l2_packed_fwd(ops._p(x), ops._p(table), ops._p(packed), ops._p(temp), ops._p(p_buf), ops._p(idx_buf, torch.int64),
                                    ops._p(out_buf), n, D, V, ops.stream_handle())

        def launch_with_pack():                    # training: the table moves every step, the pack launch precedes every search
            lib.st_vq_l2_fwd(ops._p(x), ops._p(table), ops._p(temp), ops._p(p_buf), ops._p(idx_buf, torch.int64),
                             ops._p(out_buf), ops._p(ws), n, D, V, ops.stream_handle())
        inner = 20
        reps = max(args.steps, 5)
        timed = {}
        for key, fn in (('with_pack', launch_with_pack), ('packed', launch)):
            gph = ops.Graph()
            fn()
            with gph.capture():
                for _ in range(inner):
                    fn()
            gph.launch()
            torch.cuda.synchronize()
            with event_timer(lib)() as tm:
                for _ in range(reps):
                    gph.launch()
            timed[key] = tm.ms * 1e3 / (reps * inner)
            assert torch.equal(idx_buf, idx)
        us = timed['packed']
        alg = n * (520 + 4 * V) + V * D * 4
        row = {'utterances': Bn, 'vectors': n, 'V': V, 'us_per_launch': round(us, 3), 'us_with_pack_launch': round(timed['with_pack'], 3),
               'algorithmic_bytes': alg,
               'GBps': round(alg / (us * 1e-6) / 1e9, 1), 'frac_of_hbm_peak': round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
               'vectors_per_s': round(n / (us * 1e-6), 1), 'GFLOPs': round(2.0 * n * V * D / (us * 1e-6) / 1e9, 1)}
        rows.append(row)
        if (Bn, V) == (32, 512):
            head = row
    if rk.rank != 0:
        return None
    cpu = None
    if not args.no_cpu_baseline and rk.world == 1:
        # the reference's own formulation on the host: neg_batch_l2 (expanded squares) -> softmax -> argmax -> embedding lookup
        # (src/embed.py:105-147,208-213; the codebook has no nn module beyond the table, so this is the functional restatement)
        from oracle import vq_oracle as VQ
        gc = torch.Generator().manual_seed(7)
        xc, tc = torch.randn(32, 129, 64, generator=gc), torch.randn(512, 64, generator=gc)
        Wc = {'learnable_table': tc, 'temp': torch.ones(1)}
        cpu = cpu_timed(lambda: VQ.l2_forward(Wc, xc[:4]), lambda i: [VQ.l2_forward(Wc, xc) for _ in range(20)], 20 * 32 * 129,
                        'vectors/s', 'oracle_port', '20 calls of L2Embedding.forward on (32,129,64) latents, V=512 (oracle/vq_oracle.py: the '
                        'reference\'s torch formulation); probe = 4 utterances', max_passes=3)
    res = {'metric': 'VQ vectors/sec (L2Embedding.forward nearest-code search)', 'value': head['vectors_per_s'] * rk.world,
            'unit': 'vectors/s', 'n_gpus': rk.world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': head['us_per_launch'] * 1e-3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'C3: L2Embedding.forward on (32,129,64) latents, V=512 synthetic table (config 3); other shapes in `cases`'},
            'roofline': {'bound': 'hbm', 'kernel': 'vq_l2_mfma_kernel (table packed once per weight version)', 'achieved': head['GBps'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': head['frac_of_hbm_peak'], **_quoted_traffic('_c3'),
                         'mfma': {'achieved': round(head['GFLOPs'] / 1e3, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                  'frac': round(head['GFLOPs'] / 1e3 / MFMA_F32_PEAK_TFLOPS, 4)},
                         'algorithmic_bytes_per_launch': head['algorithmic_bytes'], 'avg_launch_us': head['us_per_launch']},
            **rk.collectives_flat(), 'cases': rows}
    if cpu is not None:
        res['cpu_baseline'] = cpu
    return res


# ===================================================================================== training (C4)
def bench_train(args, rk):
    """BASELINE config 4: config/semi-multi-spkr-paired-data.yaml, utterance-level data parallelism -- every rank runs the
    paired TTS training step on its own B=32 synthetic batch (109 speakers), BatchNorm statistics over the global batch
    (SyncBN), gradients summed by RCCL all-reduce, identical clip + Adam on every rank.
    ms_allreduce / ms_syncbn = step time with the collective minus step time without it (exposed cost)."""
    import torch
    import yaml
    from argparse import Namespace
    from semi_tts_amd import parallel
    from semi_tts_amd.solver import TtsTrainer
    config = yaml.safe_load(open(os.path.join(REPO, 'config', 'semi-multi-spkr-paired-data.yaml')))
    paras = Namespace(batch_size=B, frames=T_RAW, n_batches=1, seed=0, verbose=False, max_step=10 ** 9, load=None, n_spkr=109)
    tr = TtsTrainer(config, paras, 'train').load_data().set_model()
    batch = [t.to(rk.dev) for t in tr.batches[0]]
    last = {}
    # no host round trip inside a step (statistics read after the timed region, NaN steps skipped on the device): the host issues
    # step k+1 while the GPU finishes step k.  --sync-stats restores the reference's per-step loss.item() / isnan(grad_norm) reads
    tr.async_stats = not args.sync_stats

    def step():
        last['st'] = tr.train_step(*batch)

    step()
    variants = {}
    elapsed = rk.timed(step, args.steps, args.warmup)
    variants['full'] = elapsed
    issue_ms = rk.issue_seconds / args.steps * 1e3
    counts = parallel.collective_counts()
    counts['async_grad_buckets'] = parallel.async_bucket_count()
    if (rk.world > 1 or rk.forced) and not args.no_variants:
        parallel.set_gradient_allreduce(False)
        variants['no_allreduce'] = rk.timed(step, args.steps, 1)
        parallel.set_gradient_allreduce(True)
        parallel.sync_batchnorm(False)
        variants['no_syncbn'] = rk.timed(step, args.steps, 1)
        parallel.set_gradient_allreduce(False)                    # the reducer attached, no collective at all: its own bookkeeping
        variants['reducer_only'] = rk.timed(step, args.steps, 1)
        parallel.set_gradient_allreduce(True)
        parallel.sync_batchnorm(True)
    # a starved in-launch hand-off poisons the forward with NaN and the guarded Adam then skips every update on the device: a step
    # time measured over skipped steps is not a measurement
    tr.check_device_status()
    assert float(last['st']['grad_norm']) == float(last['st']['grad_norm']), 'non-finite gradient norm in the timed steps'
    if rk.rank != 0:
        return None
    cpu = None
    if not args.no_cpu_baseline and rk.world == 1:
        cpu = cpu_baseline_train(tr, batch, config)
    ms = {k: v / args.steps * 1e3 for k, v in variants.items()}
    frames = rk.world * batch[2].shape[0] * batch[2].shape[1] * args.steps
    n_par = sum(p.numel() for p in tr.model.parameters() if p.requires_grad)
    return {'metric': 'training mel-frames/sec (paired TTS step: fwd + loss + bwd + all-reduce + clip + Adam)',
            'value': round(frames / elapsed, 1), 'unit': 'mel-frames/s', 'n_gpus': rk.world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms['full'], 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'C4: TtsTrainer.train_step, B=%d per GPU, %d->%d frames, L=%d, 109 speakers, tf_rate=1, '
                                   'config/semi-multi-spkr-paired-data.yaml' % (B, T_RAW, batch[2].shape[1], batch[0].shape[1]),
                       'parallelism': 'dp%d (utterance-sharded, SyncBN, gradient all-reduce of %.1f MB)' % (rk.world, n_par * 4 / 1e6)},
            **rk.collectives_flat(),
            'ms_allreduce': round(ms['full'] - ms['no_allreduce'], 3) if 'no_allreduce' in ms else 0.0,
            'ms_syncbn': round(ms['full'] - ms['no_syncbn'], 3) if 'no_syncbn' in ms else 0.0,
            'ms_variants': {k: round(v, 3) for k, v in ms.items()},
            'collectives_per_step': counts,
            **({'reducer': dict(tr.reducer.stats, buckets=len(tr.reducer.buckets), static_graph=tr.reducer._sparse)} if getattr(tr, 'reducer', None) is not None else {}),
            # whole-step roofline: forward 132 GFLOP per C2 batch, training ~3x that, against the fp32 matrix peak
            'roofline': {'bound': 'mfma', 'kernel': 'whole training step (%s; the largest shares are the per-step products of the two loops and the weight-gradient GEMM tn_dma_kernel)' % _train_launches(),
                         'achieved': round(3 * 132.4e9 / (ms['full'] * 1e-3) / 1e12, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(3 * 132.4e9 / (ms['full'] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), 'traffic': None},
            'ms_host_issue_per_step': round(issue_ms, 3),      # (the host's time to issue a step of the 'full' variant; close to ms_per_step = the step is bound by the host)
            'mt_table_misses': int(__import__('semi_tts_amd._lib', fromlist=['load']).load().st_mt_table_misses()),      # (block maps of the multi-tensor launches built so far: once per set of tensor addresses)
            'last': {k: float(last['st'][k]) for k in ('loss', 'grad_norm')}, 'stats_read': 'per step' if args.sync_stats else 'after the timed steps', 'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            **({'cpu_baseline': cpu} if cpu is not None else {})}


# ===================================================================================== the two cycles (C3 as the reference trains it)
def cycle_flops(Bp, Bu, T, L, Ta, Tau, dec, enc_asr, n_mels=N_MELS):
    """algorithmic FLOPs of one cycle step's FORWARD: CTC speech encoder on the paired (+ unpaired) aug_mel, VQ search, the TTS branch
    (SURVEY 8d per-unit figures: decoder step, text encoder per position, CBHG + Linear per frame) at the batch / text length it ran at"""
    dim, k, st = enc_asr['dim'], enc_asr['kernel'], enc_asr['stride']
    H = enc_asr['rnn_dim']
    Tm = max(Ta, Tau) if Bu else Ta
    cin, t, asr = n_mels, Tm, 0.0
    for kk, ss in zip(k, st):
        t = (t + 2 * (1 if kk != 1 else 0) - kk) // ss + 1
        asr += 2.0 * cin * kk * dim * t
        cin = dim
    for layer in range(enc_asr['rnn_layers']):
        asr += 2.0 * (cin + H) * 4 * H * 2 * t
        cin = 2 * H
    asr += 2.0 * cin * 64 * t + 2.0 * 64 * 43 * t
    Bt = Bp + Bu
    r = dec['n_frames_per_step']
    steps = T // r
    tts = Bt * (steps * 2.0 * (18847232 + 10944 * L) + 2.0 * (4358144 + 131072) * L + 1.655e6 * T)
    return asr * Bt + tts


def bench_cycle(args, rk):
    """BASELINE config 3 as the reference trains it (main.py:61-63 -> bin/train_vqvae.py:111-270): VqvaeTrainer.exec's alternation of the
    speech -> text -> speech cycle (even steps; with the unpaired batch: speech encoder + VQ search + run-length merge on paired ||
    unpaired aug_mel, then the TTS branch on paired text || merged unpaired latents, unpair_speech_weight 10) and the
    text -> speech -> text cycle (odd steps; unpair_text_weight 0: the paired batch alone through the TTS branch, then the speech encoder).
    Timed: K steps of the alternating loop, then K steps of each kind on its own (kinds.*)."""
    import torch
    import yaml
    from argparse import Namespace
    from semi_tts_amd import parallel
    from semi_tts_amd.solver import VqvaeTrainer
    config = yaml.safe_load(open(os.path.join(REPO, 'config', 'semi-single-spkr-paired-data.yaml')))
    Bc = args.batch_size or B
    paras = Namespace(batch_size=Bc, unpair_batch_size=args.unpair_batch_size or Bc, frames=T_RAW, n_batches=1, seed=0, verbose=False,
                      max_step=10 ** 9, load=None, n_spkr=109)
    tr = VqvaeTrainer(config, paras, 'train').load_data().set_model()
    tr.async_stats = not args.sync_stats
    pair, unpair = tr.fetch_data('pair_iter'), tr.fetch_data('unpair_iter')
    last = {}

    def loop_step():
        kind, use = tr.cycle_kind(tr.step)
        last[kind] = tr.cycle_step(pair, unpair if use else None)

    def only(kind):
        def f():
            tr.step += (tr.step % 2) != (0 if kind == 'speech_first' else 1)      # (the step counter picks the cycle; the lr schedule moves by one step)
            last[kind] = tr.cycle_step(pair, unpair if tr.cycle_kind(tr.step)[1] else None)
        return f

    tr.step = 2                                   # past unpair_speech_start_step (0): the unpaired batch joins from the first timed step on
    loop_step(); loop_step()
    steps = args.steps + (args.steps % 2)         # whole pairs of cycles
    elapsed = rk.timed(loop_step, steps, args.warmup + (args.warmup % 2))
    issue_ms = rk.issue_seconds / steps * 1e3
    kinds = {}
    for kind in ('speech_first', 'text_first'):
        e = rk.timed(only(kind), args.steps, 2)
        kinds[kind] = {'ms_per_step': round(e / args.steps * 1e3, 3), 'ms_host_issue_per_step': round(rk.issue_seconds / args.steps * 1e3, 3)}
    tr.drain_stats()
    tr.check_device_status()
    for kind in kinds:
        assert float(last[kind]['grad_norm']) == float(last[kind]['grad_norm']), 'non-finite gradient norm in the timed steps (%s)' % kind
    if rk.rank != 0:
        return None
    mel, aug, _, text, _ = pair
    umel, uaug = unpair[0], unpair[1]
    Bp, Tp, Lp = mel.shape[0], mel.shape[1], text.shape[1]
    Bu = umel.shape[0]
    Lu = int(last['speech_first'].get('unpair_text_len', 0))
    dec = config['model']['decoder']['decoder']
    enc = config['model']['encoder']
    fl = {'speech_first': 3 * cycle_flops(Bp, Bu if Lu else 0, max(Tp, umel.shape[1]), max(Lp, Lu), aug.shape[1], uaug.shape[1], dec, enc),
          'text_first': 3 * cycle_flops(Bp, 0, Tp, Lp, aug.shape[1], 0, dec, enc)}
    frames = {'speech_first': Bp * Tp + (Bu * umel.shape[1] if Lu else 0), 'text_first': Bp * Tp}
    for kind, d in kinds.items():
        tf = fl[kind] / (d['ms_per_step'] * 1e-3) / 1e12
        d.update(mel_frames_per_s=round(frames[kind] / (d['ms_per_step'] * 1e-3), 1), gflop_per_step=round(fl[kind] / 1e9, 1),
                 roofline={'bound': 'mfma', 'achieved': round(tf, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                           'frac': round(tf / MFMA_F32_PEAK_TFLOPS, 4), 'traffic': None},
                 last={k: float(last[kind][k]) for k in ('loss', 'grad_norm', 'asr_loss', 'tts_loss') if k in last[kind]})
    kinds['speech_first']['unpair_text_len'] = Lu
    total_frames = rk.world * (frames['speech_first'] + frames['text_first']) * (steps // 2)
    total_fl = (fl['speech_first'] + fl['text_first']) * (steps // 2)
    tf = total_fl / elapsed / 1e12
    cpu = None
    if not args.no_cpu_baseline and rk.world == 1:
        cpu = cpu_baseline_cycle(tr, pair, unpair, config)
    n_par = sum(p.numel() for p in tr.model.parameters() if p.requires_grad)
    return {'metric': 'training mel-frames/sec (VqvaeTrainer cycles: speech_to_text + VQ + mean_forward + text_to_speech, CTC + freq losses, bwd, clip, Adam)',
            'value': round(total_frames / elapsed, 1), 'unit': 'mel-frames/s', 'n_gpus': rk.world, 'steps': steps,
            'warmup': args.warmup + (args.warmup % 2), 'ms_per_step': round(elapsed / steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'C3 cycles: VqvaeTrainer.exec alternation (even steps speech-first with the unpaired batch, odd steps text-first), '
                                   'paired B=%d + unpaired B=%d per GPU, %d->%d frames, aug_mel %d frames, paired L=%d, merged unpaired L=%d, '
                                   'L2 codebook V=%d, config/semi-single-spkr-paired-data.yaml'
                                   % (Bp, Bu, T_RAW, Tp, aug.shape[1], Lp, Lu, tr.vocab_size),
                       'parallelism': 'dp%d (utterance-sharded, SyncBN, gradient all-reduce of %.1f MB)' % (rk.world, n_par * 4 / 1e6)},
            **rk.collectives_flat(),
            'roofline': {'bound': 'mfma', 'kernel': 'whole alternating step (3 x forward FLOPs of the speech encoder, VQ search and TTS branch at the shapes they ran at)',
                         'achieved': round(tf, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tf / MFMA_F32_PEAK_TFLOPS, 4), 'traffic': None},
            'kinds': kinds, 'ms_host_issue_per_step': round(issue_ms, 3),
            'stats_read': 'per step' if args.sync_stats else 'after the timed steps (the merged lengths of mean_forward are the one host read of a speech-first step)',
            'collectives_per_step': parallel.collective_counts(), 'ctc_nan': int(getattr(tr, 'ctc_nan', 0)),
            'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            **({'cpu_baseline': cpu} if cpu is not None else {})}


def cpu_baseline_cycle(tr, pair, unpair, config):
    """both cycles on the host through the torch.nn assembly of the whole VQVAE (oracle/nn_baseline.py: NNVqvae) under torch autograd:
    one speech-first step with the unpaired batch + one text-first step = one unit of the alternating loop"""
    import torch
    from oracle import nn_baseline as NB
    from oracle import tts_oracle as O
    W = {k: v.detach().cpu() for k, v in tr.model.state_dict().items()}
    net = NB.NNVqvae(W, config['model'], tr.n_mels).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    sr = config['data']['audio']['sample_rate']
    fl = lambda p, l: O.freq_loss(p, l, sr, tr.n_mels)
    pc, uc = tuple(t.cpu() for t in pair), tuple(t.cpu() for t in unpair)
    cut = lambda b, n, f: (b[0][:n, :f], b[1][:n, :f], b[2][:n, :f], b[3][:n], b[4][:n])
    hp = config['hparas']

    def both(p, u):
        NB.cycle_step(net, opt, 'speech_first', p, u, fl, hp)
        NB.cycle_step(net, opt, 'text_first', p, None, fl, hp)
    short = lambda: both(cut(pc, 2, 24), cut(uc, 2, 24))
    full = lambda i: both(pc, uc)
    frames = pc[0].shape[0] * pc[0].shape[1] * 2 + uc[0].shape[0] * uc[0].shape[1]
    return cpu_timed(short, full, frames, 'mel-frames/s', 'nn_modules_autograd',
                     'one speech-first step (paired B=%d || unpaired B=%d) + one text-first step (paired) of the torch.nn assembly of the whole VQVAE '
                     '(oracle/nn_baseline.py: NNCtc + L2 codebook + host mean_forward + NNTacotron2, CTC + freq losses, backward, clip 5.0, Adam) '
                     'under torch autograd; probe = the same at B=2+2, 24 frames' % (pc[0].shape[0], uc[0].shape[0]),
                     max_passes=2, budget=(30.0, 50.0))


def _train_launches():
    """launches of one training step as counted by the committed rocprofv3 kernel trace of one step (tools/gpu_train_prof.sh)"""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_train_one_step_kernel_stats.csv')))
    if not files:
        return 'launch count: no committed kernel trace'
    n = sum(float(r['launches_per_step']) for r in csv.DictReader(open(files[-1])))
    return '%d launches per step in profiles/%s' % (round(n), os.path.basename(files[-1]))


def cpu_baseline_train(tr, batch, config):
    """the paired TTS training step on the host: forward (teacher forcing) + freq_loss + backward + clip + Adam through the
    torch.nn assembly of Tacotron2 (oracle/nn_baseline.py) and torch autograd -- what the reference's step executes on CPU"""
    import torch
    from oracle import nn_baseline as NB
    from oracle import tts_oracle as O
    text, sid, mel, linear = batch
    with torch.no_grad():
        txt_embed = tr.model.codebook.inference(text).cpu()
        spk = tr.model.embed_speakers(sid).cpu()
    W = {k[4:]: v.detach().cpu() for k, v in tr.model.state_dict().items() if k.startswith('tts.')}
    hp = dict(config['model']['decoder']['decoder'])
    hp['n_mels'] = tr.n_mels
    net = NB.NNTacotron2(W, hp).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    mel_c, lin_c = mel.cpu(), linear.cpu()
    sr = config['data']['audio']['sample_rate']
    fl = lambda p, l: O.freq_loss(p, l, sr, tr.n_mels)
    short = lambda: NB.train_step(net, opt, txt_embed[:4], spk[:4], mel_c[:4, :24], lin_c[:4, :24], fl)
    full = lambda i: NB.train_step(net, opt, txt_embed, spk, mel_c, lin_c, fl)
    Bn, Tn = mel_c.shape[0], mel_c.shape[1]
    return cpu_timed(short, full, Bn * Tn, 'mel-frames/s', 'nn_modules_autograd',
                     'full training steps (B=%d, %d frames: teacher-forced forward + freq_loss + backward + clip 5.0 + Adam) of the torch.nn '
                     'assembly of Tacotron2 (oracle/nn_baseline.py) under torch autograd; probe = one step at B=4, 24 frames' % (Bn, Tn),
                     max_passes=2, budget=(30.0, 50.0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-graph', action='store_true', help='issue the decode loop eagerly instead of replaying a hipGraph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--sync-stats', action='store_true', help='--workload train: read loss / grad norm on the host every step (as the reference does)')
    ap.add_argument('--pre-parts', type=int, default=0, help=argparse.SUPPRESS)      # experiment: workgroups per utterance of the attention pre part
    ap.add_argument('--vq-head-only', action='store_true', help='c3: only the headline case (32 x 129 vectors, V = 512): PMC passes')
    ap.add_argument('--no-variants', action='store_true', help='--workload train: only the full step (no timed variants without the collectives): for kernel traces')
    ap.add_argument('--dist', action='store_true',
                    help='with --gpus 1: initialise a world-size-1 process group (RCCL) and issue every collective anyway')
    ap.add_argument('--no-finite-check', action='store_true', help=argparse.SUPPRESS)    # timing experiments (tools/gpu_ablate.sh)
    ap.add_argument('--traffic-json', default=None, help='PMC summary (tools/pmc_summary.py) to quote as roofline.traffic')
    ap.add_argument('--batch-size', type=int, default=None, help='--workload cycle: paired utterances per GPU (default 32; the config file says 8)')
    ap.add_argument('--unpair-batch-size', type=int, default=None, help='--workload cycle: unpaired utterances per GPU (default: --batch-size)')
    ap.add_argument('--workload', choices=['c2', 'c5', 'c3', 'train', 'cycle'], default='c2',
                    help="c2 = the headline configuration; c5 / c3 / train = secondary lines (see the module docstring)")
    ap.add_argument('--rccl-probe', action='store_true', help=argparse.SUPPRESS)        # child of a rank: see Ranks._probe_rccl
    args = ap.parse_args()
    if args.rccl_probe:
        return rccl_probe_main()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))            # nothing above this line has touched the GPU

    # The contract is ONE JSON line on stdout.  Libraries print there too (RCCL writes its version banner to stdout when the first
    # communicator comes up): from here on file descriptor 1 is stderr, and the result line alone goes to the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rk = Ranks(args)
    fn = {'c2': bench_decode, 'c5': bench_decode, 'c3': bench_vq, 'train': bench_train, 'cycle': bench_cycle}[args.workload]
    res = fn(args, rk)
    if rk.rank == 0:
        os.write(real_stdout, (json.dumps(res) + '\n').encode())
    rk.close()


if __name__ == '__main__':
    main()
