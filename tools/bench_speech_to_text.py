#!/usr/bin/env python3
"""Secondary line: VQVAE.speech_to_text forward (CTC speech encoder -> L2 codebook search -> run-length merge) at config-3 shapes,
eval mode, eager and as ONE hipGraph replay (runtime.GraphedSpeechToText), plus the training-mode forward (dropout, batch statistics).

    python tools/bench_speech_to_text.py [--batch-size 32 --unpair-batch-size 32 --frames 256 --steps 20]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch   # noqa: E402
import yaml    # noqa: E402


def timed(fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch-size', type=int, default=32)
    ap.add_argument('--unpair-batch-size', type=int, default=32)
    ap.add_argument('--frames', type=int, default=256)
    ap.add_argument('--steps', type=int, default=20)
    a = ap.parse_args()
    from semi_tts_amd.runtime import GraphedSpeechToText
    from semi_tts_amd.synthetic import load_synthetic, synthetic_cycle_batch
    from semi_tts_amd.vqvae import VQVAE
    cfg = yaml.safe_load(open(os.path.join(ROOT, 'config', 'semi-single-spkr-paired-data.yaml')))
    mc = cfg['model']
    mc['codebook'].update(phn_attr_pth='', proj_attr=None)
    dev = torch.device('cuda', 0)
    m = VQVAE(80, 1025, 43, 109, **mc)
    load_synthetic(m, 1234)
    m = m.to(dev).eval()
    pair = [t.to(dev) for t in synthetic_cycle_batch(a.batch_size, a.frames, 3, seed=1)]
    un = [t.to(dev) for t in synthetic_cycle_batch(a.unpair_batch_size, a.frames, 3, seed=2)]
    out = {}
    with torch.no_grad():
        out['ms_eval_eager'] = timed(lambda: m.speech_to_text(pair[1], un[1]), a.steps)
    gs = GraphedSpeechToText(m, a.batch_size, a.frames, dev, B_unpair=a.unpair_batch_size).capture()
    gs.mel[:a.batch_size].copy_(pair[1]); gs.mel[a.batch_size:].copy_(un[1])
    out['ms_eval_graph_replay'] = timed(lambda: gs.graph.launch(), a.steps)
    m.train()
    with torch.no_grad():
        out['ms_train_mode_no_grad'] = timed(lambda: m.speech_to_text(pair[1], un[1]), a.steps)
    out['ms_train_mode_autograd_fwd'] = timed(lambda: m.speech_to_text(pair[1], un[1]), a.steps)
    # forward + backward of the half on its own: random gradients into everything the cycle reads (posteriors of both parts, paired latents,
    # merged unpaired latents); parameter gradients are dropped after each pass
    g = torch.Generator(device=dev).manual_seed(3)
    r = lambda t: torch.rand(t.shape, device=dev, generator=g)

    def fwd_bwd():
        pp, pl, up, ul, _, _, _ = m.speech_to_text(pair[1], un[1])
        outs = [t for t in (pp, pl, up, ul) if t is not None and t.requires_grad]
        torch.autograd.backward(outs, [r(t) for t in outs])
        for p_ in m.parameters():
            p_.grad = None
    out['ms_train_fwd_bwd'] = timed(fwd_bwd, a.steps)
    Bt, Ta = a.batch_size + a.unpair_batch_size, a.frames
    sys.path.insert(0, ROOT)
    from bench import cycle_flops, MFMA_F32_PEAK_TFLOPS
    enc = cfg['model']['encoder']
    # (the speech-encoder + VQ share of cycle_flops: the TTS terms vanish at T = 0, L = 0)
    fl = cycle_flops(a.batch_size, a.unpair_batch_size, 0, 0, Ta, Ta, cfg['model']['decoder']['decoder'], enc)
    best = out['ms_eval_graph_replay']
    tf = fl / (best * 1e-3) / 1e12
    print(json.dumps(dict(metric='speech_to_text forward: aug-mel frames/sec (CTC speech encoder + VQ search + run-length merge)',
                          value=round(Bt * Ta / (best * 1e-3), 1), unit='mel-frames/s', n_gpus=1, steps=a.steps, dtype='f32', data='synthetic',
                          config=dict(workload='VQVAE.speech_to_text, paired B=%d || unpaired B=%d, %d aug-mel frames -> %d positions, V=43'
                                      % (a.batch_size, a.unpair_batch_size, Ta, Ta // 2)),
                          gflop=round(fl / 1e9, 2),
                          roofline=dict(bound='mfma', achieved=round(tf, 2), peak=MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s',
                                        frac=round(tf / MFMA_F32_PEAK_TFLOPS, 4), traffic=None),
                          **{k: round(v, 4) for k, v in out.items()})))


if __name__ == '__main__':
    main()
