#!/usr/bin/env python
"""Entry point with the reference's flags (ref: main.py:14-68): YAML config + Solver dispatch.
The dispatch is the reference's (main.py:49-63): `--gen-specgram` runs batched free-running synthesis
(bin/gen_specgram.py), the default mode runs `VqvaeTrainer` -- the alternating speech -> text -> speech /
text -> speech -> text cycles of bin/train_vqvae.py:111-270 (CTC speech encoder, codebook, run-length merge,
TTS branch, CTC + freq losses, backward, clip + Adam, all on the HIP kernels) -- on synthetic paired and
unpaired batches (the corpus is not available).  `--tts-only` keeps the paired TTS step alone (TtsTrainer).

    python main.py --config config/semi-single-spkr-paired-data.yaml --max-step 20 [--frames 256 --batch-size 8]
    python main.py --config config/supervised.yaml --gen-specgram [--load ckpt.pth] [--frames 256 --batch-size 32]
    python main.py --config config/supervised.yaml --tts-only --max-step 20 [--frames 256 --batch-size 32]
"""
import argparse
import random

import numpy as np
import torch
import yaml

parser = argparse.ArgumentParser(description='semi-tts decode path on MI355X.')
parser.add_argument('--config', type=str, help='Path to experiment config.')
parser.add_argument('--name', default=None, type=str, help='Name for logging.')
parser.add_argument('--logdir', default='log/', type=str, help='Logging path.')
parser.add_argument('--ckpdir', default='ckpt/', type=str, help='Checkpoint/Result path.')
parser.add_argument('--load', default=None, type=str, help='Load pre-trained model')
parser.add_argument('--seed', default=0, type=int, help='Random seed for reproducable results.')
parser.add_argument('--njobs', default=5, type=int, help='(unused: no data loader workers)')
parser.add_argument('--cpu', action='store_true', help='Not supported: the path is MI355X-only.')
parser.add_argument('--debug', action='store_true', help='Debug use. (parsed and never read by the reference either)')
parser.add_argument('--no-pin', action='store_true', help='Disable pin-memory for dataloader (no data loader here: ignored)')
parser.add_argument('--asr-decode', action='store_true', help='ASR beam decode (the reference dispatches to bin/asr_decode.py, which its tree lacks)')
parser.add_argument('--gen-specgram', action='store_true', help='Generating mel/linear spectrogram.')
parser.add_argument('--gen-gt-specgram', action='store_true', help='(the reference dispatches to bin/gen_gt_specgram.py, which its tree lacks)')
parser.add_argument('--no-msg', action='store_true', help='Hide all messages.')
parser.add_argument('--actual-len', action='store_true', help='Using actual len for CTC loss. (synthetic inputs are full length: no effect)')
parser.add_argument('--store-best-per', action='store_true', help='Only store the model with best PER. (no dev corpus: no effect)')
parser.add_argument('--asr-only', action='store_true', help='(the reference dispatches to bin/train_asr.py, which its tree lacks)')
parser.add_argument('--gen-wav', action='store_true', help='Generate waveform using Griffin-Lim. (audio back end out of scope: ignored)')
# synthetic-data knobs (the reference reads these from the corpus)
parser.add_argument('--frames', default=256, type=int, help='mel frames per synthetic utterance')
parser.add_argument('--batch-size', default=None, type=int)
parser.add_argument('--n-batches', default=1, type=int)
parser.add_argument('--unpair-batch-size', default=None, type=int, help='utterances per synthetic unpaired batch (default: --batch-size)')
parser.add_argument('--unpair-frames', default=None, type=int, help='mel frames per synthetic unpaired utterance (default: --frames)')
parser.add_argument('--stretch', action='store_true', help='aug_mel lengths drawn from data.audio.time_stretch_range (default: unstretched)')
parser.add_argument('--tts-only', action='store_true', help='train the paired TTS branch alone (TtsTrainer) instead of the two cycles')
parser.add_argument('--max-step', default=None, type=int, help='training steps (default: hparas.max_step)')
parser.add_argument('--save', action='store_true', help='write ckpt/<name>/latest.pth ({model, optimizer, global_step}) after training')
parser.add_argument('--async-stats', action='store_true', help='training: no host read of loss / gradient norm inside a step (read when logged; '
                    'a NaN gradient norm skips the update on the device)')


# Flags of the reference (main.py:23-33) with nothing to do on this path, and those whose solver files are
# absent from the reference tree itself (main.py:49-60 import bin/asr_decode.py, bin/gen_gt_specgram.py,
# bin/train_asr.py, none of which exist): accepted by the parser so that existing launch lines keep working.
IGNORED_FLAGS = ('debug', 'no_pin', 'actual_len', 'store_best_per', 'gen_wav')
MISSING_SOLVERS = {'asr_decode': 'bin/asr_decode.py', 'gen_gt_specgram': 'bin/gen_gt_specgram.py',
                   'asr_only': 'bin/train_asr.py'}


def parse_args(argv=None):
    """The reference's command lines parse unchanged (ref: main.py:14-41); returns the same `paras`
    attributes it sets (`gpu`, `pin_memory` -- inverted there as here, main.py:40 --, `verbose`)."""
    paras = parser.parse_args(argv)
    setattr(paras, 'gpu', not paras.cpu)
    setattr(paras, 'pin_memory', False if paras.cpu else paras.no_pin)
    setattr(paras, 'verbose', not paras.no_msg)
    for flag, path in MISSING_SOLVERS.items():
        if getattr(paras, flag):
            parser.error('--%s: the reference dispatches this mode to %s, which is not part of the reference tree; '
                         'only the default (training) and --gen-specgram modes exist' % (flag.replace('_', '-'), path))
    if paras.verbose:
        for flag in IGNORED_FLAGS:
            if getattr(paras, flag):
                print('[INFO] --%s accepted for compatibility; it has no effect on this path' % flag.replace('_', '-'))
    return paras


def main(argv=None):
    paras = parse_args(argv)
    config = yaml.load(open(paras.config, 'r'), Loader=yaml.FullLoader)
    if paras.batch_size is None:
        paras.batch_size = config['data']['corpus'].get('batch_size', 8)
    random.seed(paras.seed)
    np.random.seed(paras.seed)
    torch.manual_seed(paras.seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(paras.seed)
    import os
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:      # one process per GPU (torch.distributed.run), RCCL over xGMI
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl')
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    if paras.gen_specgram:
        from semi_tts_amd.solver import SpecgramGenerator as Solver
        mode = 'test'
    elif paras.tts_only:
        from semi_tts_amd.solver import TtsTrainer as Solver
        mode = 'train'
    else:                                                    # ref: main.py:61-63
        from semi_tts_amd.solver import VqvaeTrainer as Solver
        mode = 'train'
    solver = Solver(config, paras, mode)
    solver.load_data()
    solver.set_model()
    solver.exec()


if __name__ == '__main__':
    main()
