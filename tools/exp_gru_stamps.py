#!/usr/bin/env python3
"""Phase stamps of one step of the CBHG BiGRU kernel (workgroup (0, 0), wave 0, step 100): needs a library built with -DGRU_STAMPS
(ST_LIB_PATH=tools/variants/libsemitts_grustamps.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semi_tts_amd import _lib, ops
lib = _lib.load()
dev = torch.device('cuda')
B, T, H = 32, 258, 80
gi = [torch.randn(B, T, 3 * H, device=dev) for _ in range(2)]
w = [torch.randn(3 * H, H, device=dev) / H ** 0.5 for _ in range(2)]
b = [torch.randn(3 * H, device=dev) * 0.1 for _ in range(2)]
out = torch.zeros(B, T, 2 * H, device=dev)
NAMES = ['h reads landed (7 ds_read_b128)', '84 multiply-adds + 3 slice merges (DPP)', 'gates + update', 'LDS write + stores issued', 'barrier', '(loop overhead to next step)']
for tape in (None, torch.zeros(2, B, T, 4, H, device=dev)):
    for _ in range(3):
        _lib.check(lib.st_gru_seq_fwd(ops._p(gi[0]), ops._p(gi[1]), ops._p(w[0]), ops._p(w[1]), ops._p(b[0]), ops._p(b[1]), ops._p(out), 2 * H,
                                      ops._p(tape), B, T, H, 2, ops.stream_handle()), 'gru')
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    lib.st_gru_debug_stamps.argtypes = [C.c_void_p]
    lib.st_gru_debug_stamps(buf)
    st = list(buf)
    print('training' if tape is not None else 'inference', 'step: %d cycles' % (st[5] - st[0]))
    for i in range(5):
        print('   %-40s %6d' % (NAMES[i], st[i + 1] - st[i]))
