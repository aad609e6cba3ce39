import os, sys, yaml, torch, ctypes as C
sys.path.insert(0, os.getcwd())
from argparse import Namespace
from semi_tts_amd import _lib
from semi_tts_amd.solver import TtsTrainer
lib = _lib.load()
cfg = yaml.safe_load(open('config/semi-single-spkr-paired-data.yaml'))
paras = Namespace(batch_size=32, frames=64, n_batches=1, seed=3, verbose=False, max_step=6, load=None)
tr = TtsTrainer(cfg, paras, 'train').load_data().set_model()
tr.async_stats = True
batch = [t.to('cuda') for t in tr.batches[0]]
for i in range(8):
    tr.train_step(*batch)
f, b = (C.c_long * 3)(), (C.c_long * 3)()
lib.st_loop_graph_stats(f, b); print(list(f), list(b))
