#!/usr/bin/env python3
"""Development tool (MI355X box): is the C2 training step bound by the host (Python issuing launches) or by the GPU?
Per step: host time until everything is issued (no sync inside the step) vs wall time with one sync at the end."""
import os
import sys
import time
from argparse import Namespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import yaml
from semi_tts_amd.solver import TtsTrainer

config = yaml.safe_load(open(os.path.join(ROOT, 'config', 'semi-single-spkr-paired-data.yaml')))
paras = Namespace(batch_size=32, frames=256, n_batches=1, seed=0, verbose=False, max_step=10 ** 9, load=None)
tr = TtsTrainer(config, paras, 'train').load_data().set_model()
text, sid, mel, linear = (t.to(tr.device) for t in tr.batches[0])
sync = torch.cuda.synchronize
for i in range(8):
    sync(); t0 = time.perf_counter()
    tf_rate = tr.optimizer.pre_step(tr.step)
    mp, lp, *_ = tr.model.text_to_speech(text, sid, None, None, None, None, mel, None, tf_rate)
    total = tr.tts_weight * (tr.freq_loss(mp, mel) + tr.freq_loss(lp, linear))
    t1 = time.perf_counter()
    total.backward()
    t2 = time.perf_counter()
    gn = tr.clip_grad_norm_(tr.model.parameters(), tr.GRAD_CLIP)
    tr.optimizer.step()
    tr.step += 1
    t3 = time.perf_counter()
    sync(); t4 = time.perf_counter()
    print('step %d: host fwd %.2f ms, bwd %.2f, opt %.2f = %.2f issued; wall %.2f ms (GPU tail after the host %.2f)' %
          (i, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0), 1e3 * (t4 - t0), 1e3 * (t4 - t3)))
