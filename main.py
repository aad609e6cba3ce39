#!/usr/bin/env python
"""Entry point with the reference's flags (ref: main.py:14-68): YAML config + Solver dispatch.
Only the solvers on the TTS path exist here: `--gen-specgram` runs batched free-running
synthesis, the default mode runs the paired TTS training step (forward, loss, backward on the HIP
kernels; clip + Adam from torch), both on synthetic inputs (the corpus is not available).

    python main.py --config config/supervised.yaml --gen-specgram [--load ckpt.pth] [--frames 256 --batch-size 32]
    python main.py --config config/supervised.yaml --max-step 20 [--frames 256 --batch-size 32]
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
parser.add_argument('--gen-specgram', action='store_true', help='Generating mel/linear spectrogram.')
parser.add_argument('--no-msg', action='store_true', help='Hide all messages.')
# synthetic-data knobs (the reference reads these from the corpus)
parser.add_argument('--frames', default=256, type=int, help='mel frames per synthetic utterance')
parser.add_argument('--batch-size', default=None, type=int)
parser.add_argument('--n-batches', default=1, type=int)
parser.add_argument('--max-step', default=None, type=int, help='training steps (default: hparas.max_step)')
parser.add_argument('--save', action='store_true', help='write ckpt/<name>/latest.pth ({model, optimizer, global_step}) after training')
parser.add_argument('--async-stats', action='store_true', help='training: no host read of loss / gradient norm inside a step (read when logged; '
                    'a NaN gradient norm skips the update on the device)')


def main():
    paras = parser.parse_args()
    setattr(paras, 'gpu', not paras.cpu)
    setattr(paras, 'verbose', not paras.no_msg)
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
    else:
        from semi_tts_amd.solver import TtsTrainer as Solver
        mode = 'train'
    solver = Solver(config, paras, mode)
    solver.load_data()
    solver.set_model()
    solver.exec()


if __name__ == '__main__':
    main()
