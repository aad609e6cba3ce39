#!/usr/bin/env python3
"""Headline benchmark: mel-frames/sec of the decode+attention region (Decoder.forward:
processed memory + AdaIN statistics + every autoregressive decode step with location-sensitive
attention), free-running inference, on BASELINE.json config 2:
B = 32 utterances per GPU, 256 -> 258 frames (86 decode steps, r = 3), L = 43, n_mels = 80, fp32,
prenet dropout 0.5 active (it never turns off in the reference), synthetic weights and inputs.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch.  Weak scaling: every rank decodes its
own batch of 32 utterances (utterances are independent; there is no data-path collective in
inference).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

import numpy as np
import torch

B, T_RAW, R, L, N_MELS = 32, 256, 3, 43, 80
T = T_RAW + (R - T_RAW % R)          # 258: the reference pads r - T % r frames (bin/train_vqvae.py:43-46)
STEPS = T // R                        # 86
MFMA_F32_PEAK_TFLOPS = 157.3          # MI355X fp32 matrix peak (256 CUs x 256 FLOP/cycle x 2.4 GHz)
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def lstm_algorithmic_bytes(Bsz, H, K):
    """bytes one LSTM-cell launch must move: weights (4H x K) + biases + x/h in + c in/out + h out"""
    return 4 * (4 * H * K + 8 * H + Bsz * K + 3 * Bsz * H)


def cpu_baseline(m, txt_mem, spk, passes=3):
    """the oracle's Decoder.forward on the host cores, same workload, full 86 steps"""
    from oracle import tts_oracle as O
    from helpers import full_hp
    W = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    # B=32 GEMMs stop scaling long before a 128-core host is full (the survey measured 8 threads);
    # use min(cores, 16) threads and say so
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    hp = full_hp(0.5)
    mem, s = txt_mem.cpu(), spk.cpu()
    drop = O.DropoutSource('rng', generator=torch.Generator().manual_seed(0))
    times = []
    budget = time.perf_counter() + 40.0          # bounded sample: stop after ~40 s whatever happens
    with torch.no_grad():
        for i in range(passes + 1):
            t0 = time.perf_counter()
            O.decoder_forward(W, mem, T, s, hp, tf_rate=0.0, training=False, drop=drop)
            times.append(time.perf_counter() - t0)
            if time.perf_counter() > budget and len(times) >= 2:
                break
    passes = len(times) - 1
    t = float(np.median(times[1:]))
    return {'value': B * T / t, 'unit': 'mel-frames/s', 'cores': cores, 'kind': 'port',
            'sample': '%d full passes of Decoder.forward (B=%d, %d steps, L=%d) after 1 warm-up, median; '
                      'torch CPU fp32, %d threads' % (passes, B, STEPS, L, cores), 'seconds_per_pass': t}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-graph', action='store_true', help='issue the decode loop eagerly instead of replaying a hipGraph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', choices=['c2', 'c5'], default='c2',
                    help="c2 = the headline configuration; c5 = BASELINE config 5 (bin/gen_specgram.py long-form: B=64, L=171, "
                         "(1026+40)//3 = 355 decode steps) as a secondary line for DESIGN.md")
    args = ap.parse_args()
    global B, L, T, STEPS
    if args.workload == 'c5':
        B, L, STEPS = 64, 171, (1026 + 40) // R
        T = STEPS * R

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # RCCL ('nccl' on ROCm).  ST_BENCH_BACKEND=gloo only exists to exercise this multi-rank path on a box with fewer GPUs
        # than ranks (the ranks then share devices; the max-over-ranks reduction goes through a CPU tensor).
        backend = os.environ.get('ST_BENCH_BACKEND', 'nccl')
        dist.init_process_group(backend, rank=rank, world_size=world)
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)

    from helpers import full_tacotron
    from semi_tts_amd import ops
    from semi_tts_amd.synthetic import synthetic_batch

    m = full_tacotron(dev, seed=1234, prenet_dropout=0.5)
    txt, spk, _ = synthetic_batch(B, L, T, seed=100 + rank)
    txt, spk = torch.from_numpy(txt).to(dev), torch.from_numpy(spk).to(dev)
    with torch.no_grad():
        memory = m.encoder(txt, None).contiguous()        # inputs of the timed region, resident in HBM
    dec = m.decoder
    from semi_tts_amd.runtime import GraphedDecoder
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
    mel = state['out'][0]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], device=dev if dist.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert bool(torch.isfinite(mel).all()), 'non-finite mel output'

    # ---- roofline of the dominant kernel (the weight-streaming LSTM cell), measured with HIP events
    # on the stream the kernel runs on: the two launches of a decode step (query LSTM K=1792, decoder
    # LSTM K=2560) alternate exactly as in the loop, so the 75.5 MB of weights cycle through the caches.
    roof = None
    if rank == 0:
        import ctypes as C
        from semi_tts_amd import _lib
        lib = _lib.load()
        Q, D, E, P = dec.query_rnn_dim, dec.dec_rnn_dim, dec.enc_embed_dim, dec.prenet_dim
        f32 = dict(device=dev, dtype=torch.float32)
        wq_ih, wd_ih = dec.query_rnn.weight_ih, dec.dec_rnn.weight_ih
        pk_q = ops.pack_weight([wq_ih, wq_ih[:, P:], dec.query_rnn.weight_hh], [P, E, Q], 4 * Q, lstm_H=Q,
                               ldws=[P + E, P + E, Q])
        pk_d = ops.pack_weight([wd_ih, wd_ih[:, E:], dec.dec_rnn.weight_hh], [E, Q, D], 4 * D, lstm_H=D,
                               ldws=[E + Q, E + Q, D])
        Kq, Kd = P + E + Q, E + Q + D                       # all multiples of 16 at the headline shape
        xq = ops.tile_rows(torch.randn(B, Kq, **f32))
        xd = ops.tile_rows(torch.randn(B, Kd, **f32))
        c_q, c_d = torch.randn(B, Q, **f32), torch.randn(B, D, **f32)
        ho, co = torch.zeros(ops.t16_floats(B, Q), **f32), torch.empty(B, Q, **f32)
        xq_v, xd_v, ho_v = ops.t16_view(xq, K=Kq), ops.t16_view(xd, K=Kd), ops.t16_view(ho, K=Q)

        def pair():
            ops.lstm_cell_packed(pk_q, xq_v, Kq, dec.query_rnn.bias_ih, dec.query_rnn.bias_hh, c_q, ho_v, co, B, Q)
            ops.lstm_cell_packed(pk_d, xd_v, Kd, dec.dec_rnn.bias_ih, dec.dec_rnn.bias_hh, c_d, ho_v, co, B, D)

        # the launches are replayed from a hipGraph (as in the decode loop) so the measurement sees
        # device time, not the Python/ctypes issue rate
        inner = 50
        g2 = ops.Graph()
        pair()
        with g2.capture():
            for _ in range(inner):
                pair()
        for _ in range(3):
            g2.launch()
        torch.cuda.synchronize()
        e0, e1 = C.c_void_p(), C.c_void_p()
        lib.st_event_create(C.byref(e0)); lib.st_event_create(C.byref(e1))
        outer = 10
        reps = inner * outer
        s = ops.stream_handle()
        lib.st_event_record(e0, s)
        for _ in range(outer):
            g2.launch()
        lib.st_event_record(e1, s)
        ms = C.c_float()
        lib.st_event_elapsed_ms(e0, e1, C.byref(ms))
        lib.st_event_destroy(e0); lib.st_event_destroy(e1)
        avg_us = ms.value * 1e3 / (2 * reps)
        alg = 0.5 * (lstm_algorithmic_bytes(B, Q, P + E + Q) + lstm_algorithmic_bytes(B, D, E + Q + D))
        achieved = alg / (avg_us * 1e-6) / 1e9
        flops = 0.5 * (2.0 * B * 4 * Q * Kq + 2.0 * B * 4 * D * Kd)
        traffic = None       # HBM bytes per launch from the PMC pass of the same command (tools/gpu_pmc.sh)
        try:
            if args.workload == 'c2':
                with open(os.path.join(REPO, 'profiles', 'r01_pmc_hbm_traffic.json')) as f:
                    traffic = json.load(f)['hbm_bytes_per_launch']
        except (OSError, KeyError, ValueError):
            pass
        roof = {'bound': 'hbm', 'kernel': 'pk_lstm_rt2_kernel<8,2,1> (fused LSTM cell on packed operands: gate GEMM + pointwise; 2 row tiles x 1 batch tile per workgroup)',
                'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                'algorithmic_bytes_per_launch': alg, 'avg_launch_us': round(avg_us, 3),
                'launches_per_step': 2 * STEPS,
                # the same launch against the fp32 matrix-core peak (at B=32 the cell sits just left of the ridge:
                # 15.7 FLOP/B against 19.7; PMC evidence: profiles/r01_pmc_mfma.*)
                'mfma': {'achieved': round(flops / (avg_us * 1e-6) / 1e12, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(flops / (avg_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}}

    if rank == 0:
        frames = world * B * T * args.steps
        res = {
            'metric': 'mel-frames/sec (decode+attn)', 'value': round(frames / elapsed, 1), 'unit': 'mel-frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s: Decoder.forward free-running inference (decode+attn), B=%d per GPU, '
                                   '%s->%d frames = %d decode steps (r=3), L=%d, n_mels=%d, prenet dropout 0.5, '
                                   'config/supervised.yaml decoder section'
                                   % (args.workload.upper(), B, '256' if args.workload == 'c2' else '1026+40', T, STEPS, L, N_MELS),
                       'batch_per_gpu': B, 'frames': T, 'decode_steps': STEPS, 'text_len': L,
                       'parallelism': 'replicas x%d (utterance-sharded, no collective)' % world,
                       'launch': 'eager' if graph is None else 'hipGraph replay of the whole decode loop'},
            'us_per_decode_step': round(elapsed / args.steps / STEPS * 1e6, 2),
            'roofline': roof,
        }
        if not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(m, memory, spk)
        try:
            res['device'] = ops.device_info()
        except Exception:
            pass
        print(json.dumps(res))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
