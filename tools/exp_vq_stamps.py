#!/usr/bin/env python3
"""Phase stamps of the VQ search's tile loop (workgroup 0, wave 0, its second tile): needs a library built with -DVQ_STAMPS
(ST_LIB_PATH=tools/variants/libsemitts_vqstamps.so).  Cycles between consecutive stamps."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semi_tts_amd import _lib, ops
lib = _lib.load()
dev = torch.device('cuda')
NAMES = ['x->LDS + barrier', '|x|^2 + A fragments', 'MFMAs + prev out store + sims + row max + barrier', 'max merge + exp + barrier', 'row sum + barrier',
         'sum merge + 1/s + p + local candidates', 'p_code stores + int-min DPP', 'barrier', 'wave merge + idx + barrier', 'code row request']
for n, V in ((33024, 512), (4128, 512), (33024, 43)):
    x = torch.randn(n, 64, device=dev); table = torch.randn(V, 64, device=dev); temp = torch.ones(1, device=dev)
    packed = ops.vq_pack_table(table)
    for _ in range(3):
        ops.vq_l2(x.view(-1, 129, 64), table, temp, packed=packed)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    lib.st_vq_debug_stamps.argtypes = [C.c_void_p]
    rc = lib.st_vq_debug_stamps(buf)
    st = list(buf)[:11]
    print('n=%d V=%d rc=%d total %d cycles per tile' % (n, V, rc, st[10] - st[0]))
    for i in range(10):
        print('   %-40s %6d' % (NAMES[i], st[i + 1] - st[i]))
