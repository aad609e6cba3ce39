#!/usr/bin/env python3
"""Secondary benchmark ("Training TTS branch fwd+bwd, tf_rate=1, mel+linear freq_loss"): the paired
TTS training step of TtsTrainer at BASELINE config 2 shapes (B=32, 256->258 frames, L=43) on one MI355X.
Prints one JSON line (mel frames/s through forward + loss + backward + clip + Adam).

    python tools/bench_train.py [--steps 5 --warmup 2 --batch-size 32 --frames 256]
"""
import argparse
import json
import os
import sys
import time
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch   # noqa: E402
import yaml    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch-size', type=int, default=32)
    ap.add_argument('--frames', type=int, default=256)
    ap.add_argument('--config', default='config/semi-single-spkr-paired-data.yaml')
    ap.add_argument('--no-fuse-pw', action='store_true', help='six launches per backward step (pointwise LSTM backward as launches of its own)')
    ap.add_argument('--no-overlap-attn', action='store_true', help='attention backward and the decoder cell product as separate launches')
    ap.add_argument('--no-pair-cells', action='store_true', help='the two LSTM cells of a teacher-forced forward step as separate launches')
    a = ap.parse_args()
    from semi_tts_amd.solver import TtsTrainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    config = yaml.safe_load(open(os.path.join(root, a.config)))
    paras = Namespace(batch_size=a.batch_size, frames=a.frames, n_batches=1, seed=0, verbose=False,
                      max_step=a.steps + a.warmup, load=None)
    tr = TtsTrainer(config, paras, 'train').load_data().set_model()
    text, sid, mel, linear = (t.to(tr.device) for t in tr.batches[0])
    if a.no_fuse_pw:
        tr.model.tts.decoder.bwd_fuse_pointwise = False
    if a.no_overlap_attn:
        tr.model.tts.decoder.bwd_overlap_attn = False
    if a.no_pair_cells:
        tr.model.tts.decoder.fwd_pair_cells = False
    phases = dict(fwd=0.0, bwd=0.0, opt=0.0)
    sync = torch.cuda.synchronize
    stats = None
    for i in range(a.warmup + a.steps):
        timed = i >= a.warmup
        sync(); t0 = time.perf_counter()
        tf_rate = tr.optimizer.pre_step(tr.step)
        mp, lp, *_ = tr.model.text_to_speech(text, sid, None, None, None, None, mel, None, tf_rate)
        from semi_tts_amd import autograd as AG
        total, = AG.scalar_combine([[tr.tts_weight, tr.tts_weight]], [tr.freq_loss(mp, mel), tr.freq_loss(lp, linear)])     # (as TtsTrainer.train_step)
        sync(); t1 = time.perf_counter()
        total.backward()
        sync(); t2 = time.perf_counter()
        gn = tr.clip_grad_norm_(tr.model.parameters(), tr.GRAD_CLIP)
        tr.optimizer.step()
        tr.step += 1
        sync(); t3 = time.perf_counter()
        if timed:
            phases['fwd'] += t1 - t0; phases['bwd'] += t2 - t1; phases['opt'] += t3 - t2
        stats = dict(loss=float(total.detach()), grad_norm=float(gn))
    tot = sum(phases.values())
    frames = mel.shape[0] * mel.shape[1] * a.steps
    print(json.dumps(dict(metric='train_mel_frames_per_sec', value=frames / tot, unit='frames/s', n_gpus=1, steps=a.steps,
                          warmup=a.warmup, ms_per_step=1e3 * tot / a.steps,
                          ms_fwd=1e3 * phases['fwd'] / a.steps, ms_bwd=1e3 * phases['bwd'] / a.steps,
                          ms_opt=1e3 * phases['opt'] / a.steps, dtype='f32', data='synthetic',
                          config=dict(workload='TtsTrainer.train_step C2: B=%d, %d->%d frames, L=%d, tf_rate=1' %
                                      (mel.shape[0], a.frames, mel.shape[1], text.shape[1])),
                          last=stats, peak_mem_GB=torch.cuda.max_memory_allocated() / 2 ** 30)))


if __name__ == '__main__':
    main()
