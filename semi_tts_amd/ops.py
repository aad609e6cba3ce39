"""Thin torch-tensor front end over the C ABI (include/semitts.h).

PyTorch is used only for device memory and stream handles: every function here takes
contiguous fp32 (or int64 index) CUDA/HIP tensors, passes raw device pointers to
libsemitts_hip.so and returns.  There is no eager/CPU fallback -- a CPU tensor raises.
"""
import os
import ctypes as C
from contextlib import contextmanager

import torch

from . import _lib
from ._lib import StSeg, StGemmEpilogue, check

ACT = {None: 0, 'none': 0, 'relu': 1, 'tanh': 2, 'sigmoid': 3}

# Debug aid (tests): buffers that are handed to the kernels UNINITIALISED because every element is written before it is read (step tapes
# of the training loops, slab workspaces) are filled with NaN first -- a read of something nobody wrote then shows up in the result.
POISON_UNINIT = False


def uninit(*shape, device, dtype=torch.float32):
    """torch.empty for a buffer whose every element the kernels write before reading it (NaN-filled under POISON_UNINIT)"""
    t = torch.empty(*shape, device=device, dtype=dtype)
    if POISON_UNINIT:
        t.fill_(float('nan'))
    return t

_stream_override = None


def zeros(*shape, device, dtype=torch.float32):
    """torch.zeros through the library's fill launch (a kernel trace of a step then shows no ATen fill; same launch count)"""
    t = torch.empty(*shape, device=device, dtype=dtype)
    if dtype == torch.float32 and t.numel() > 0:
        return fill_(t, 0.0)
    return t.zero_()


def stream_handle():
    """hipStream_t (as int) all st_* calls are issued on: torch's current stream unless overridden."""
    if _stream_override is not None:
        return _stream_override
    # (the raw handle straight from the C side: torch.cuda.current_stream() builds a Stream object through four Python layers -- 10 us a
    # call, ~60 calls per training step and thread, once the host's time per step mattered: DESIGN 3.4)
    return _raw_stream(_cur_device())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)
if _raw_stream is None or _cur_device is None:                 # (an older torch: the public way)
    _raw_stream = lambda _dev: torch.cuda.current_stream().cuda_stream
    _cur_device = lambda: 0


def capturing():
    """st_* calls are being issued on an overridden stream (a hipGraph capture): nothing may read results back now"""
    return _stream_override is not None


class Starved(RuntimeError):
    """a launch whose workgroups hand data to each other was scheduled without all of them resident (a shared or CU-masked GPU):
    its outputs are NaN.  Callers that can, fall back to the multi-launch form and run again (degrade())."""


_DEGRADED = set()


def degrade(what, detail):
    """log -- once per kind -- that an in-launch hand-off was starved and its multi-launch form takes over for the rest of the process"""
    if what not in _DEGRADED:
        _DEGRADED.add(what)
        import warnings
        warnings.warn('semi_tts_amd: %s: the launch was starved of compute units (another tenant or a CU mask on this GPU); falling back '
                      'to %s for the rest of this process' % (what, detail), RuntimeWarning, stacklevel=3)


def handoff_starved(status):
    """True (and the word cleared) if an in-launch hand-off of the decode loop timed out since the word was last cleared.  Reads one
    device word (synchronises)."""
    if status is None:
        return False
    v = int(status.item())
    if v != 0:
        status.zero_()
    return v != 0


def check_handoff(status):
    """Raise if an in-launch hand-off of the decode loop timed out since the word was last cleared (st_decoder_io.handoff_status):
    the waiting workgroups were not co-resident with their producers -- the outputs of that forward are poisoned with NaN.
    Reads one device word (synchronises) and clears it."""
    if status is None:
        return
    v = int(status.item())
    if v != 0:
        status.zero_()
        raise Starved('decode loop: an in-launch hand-off of the processed query timed out (status word 0x%x): the launch was '
                           'starved of compute units; the outputs of this forward are invalid.  Set decoder.attn_pq_in_fin = False '
                           'to run the query projection and the attention as two launches.' % v)


@contextmanager
def use_stream(handle):
    global _stream_override
    prev, _stream_override = _stream_override, handle
    try:
        yield
    finally:
        _stream_override = prev


_SIDE_STREAMS = {}
_SIDE_PENDING = []


def side_stream(device):
    """a second HIP stream per device for work that is independent of the main chain (the detached postnet branch of a training step)"""
    key = str(device)
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return s


def side_pending(event=None):
    """announce (event) / clear (None) side-stream work in flight: kernels that need every compute unit to themselves -- the one-launch
    BiLSTM layers, whose workgroups wait for each other inside the launch -- make their stream wait for it first (join_side)"""
    del _SIDE_PENDING[:]
    if event is not None:
        _SIDE_PENDING.append(event)


def join_side():
    for ev in _SIDE_PENDING:
        torch.cuda.current_stream().wait_event(ev)


def _p(t, dtype=torch.float32):
    """device pointer of a tensor (None -> NULL) after checking it is usable by the kernels"""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('semi_tts_amd: the HIP path needs device tensors (got a CPU tensor); there is no CPU fallback')
    if t.dtype != dtype:
        raise RuntimeError('semi_tts_amd: expected %s, got %s' % (dtype, t.dtype))
    return t.data_ptr()


def _seg(x, w, k=None, ldx=None, ldw=None):
    s = StSeg()
    s.x, s.w = _p(x), _p(w)
    s.k = int(k if k is not None else w.shape[-1])
    s.ldx = int(ldx if ldx is not None else x.stride(-2) if x.dim() > 1 else s.k)
    s.ldw = int(ldw if ldw is not None else w.stride(-2) if w.dim() > 1 else s.k)
    return s


def _segs(pairs):
    arr = (StSeg * len(pairs))()
    for i, s in enumerate(pairs):
        arr[i] = s
    return arr


# ---------------------------------------------------------------------------------------------
def lstm_cell(segs, b_ih, b_hh, c_prev, h_out, c_out, mask=None, pre=None, ldpre=0, gates_out=None,
              ldh=None, ldc=None, ldc_prev=None):
    """segs: list of StSeg (see seg()).  h_out/c_out (B,H) views; writes in place."""
    lib = _lib.load()
    B, H = h_out.shape[0], h_out.shape[-1]
    arr = _segs(segs)
    check(lib.st_lstm_cell_fwd(arr, len(segs), _p(b_ih), _p(b_hh), _p(pre), int(ldpre),
                               _p(c_prev), int(ldc_prev if ldc_prev is not None else (c_prev.stride(0) if c_prev is not None else H)),
                               _p(mask), _p(h_out), int(ldh if ldh is not None else h_out.stride(0)),
                               _p(c_out), int(ldc if ldc is not None else c_out.stride(0)),
                               _p(gates_out), B, H, stream_handle()), 'st_lstm_cell_fwd')


seg = _seg


def skinny_linear(segs, y, bias=None, act=None, mask=None, n_split=0, y2=None, rep=0, N=None):
    lib = _lib.load()
    B = y.shape[0]
    N = int(N if N is not None else y.shape[-1])
    arr = _segs(segs)
    check(lib.st_skinny_linear_fwd(arr, len(segs), _p(bias), ACT[act], _p(mask), int(mask.stride(0)) if mask is not None else 0,
                                   _p(y), int(y.stride(0)), int(n_split), _p(y2), int(y2.stride(0)) if y2 is not None else 0,
                                   int(rep), B, N, stream_handle()), 'st_skinny_linear_fwd')
    return y


def linear_small(x, w, bias=None, act=None, mask=None, out=None):
    """y = act(x W^T + b) * mask for a small number of rows (B <= a few dozen)."""
    if out is None:
        out = torch.empty(x.shape[0], w.shape[0], device=x.device, dtype=torch.float32)
    return skinny_linear([_seg(x, w)], out, bias, act, mask)


def attn_step(pq, pm, memory, w_prev, w_cum_prev, w_out, w_cum_out, loc_conv_w, loc_lin_w, v, ctx,
              h_q=None, ada_std=None, ada_mean=None, h_adapt=None):
    lib = _lib.load()
    B, L, E = memory.shape
    A = pm.shape[-1]
    F_, _, K = loc_conv_w.shape
    Q = h_q.shape[-1] if h_q is not None else 0
    check(lib.st_attn_step_fwd(_p(pq), _p(pm), _p(memory), _p(w_prev), int(w_prev.stride(0)), _p(w_cum_prev),
                               _p(w_out), int(w_out.stride(0)), _p(w_cum_out), _p(loc_conv_w), _p(loc_lin_w), _p(v),
                               _p(ctx), int(ctx.stride(0)), _p(h_q), int(h_q.stride(0)) if h_q is not None else 0,
                               _p(ada_std), _p(ada_mean), _p(h_adapt), Q, B, L, A, E, F_, K, stream_handle()),
          'st_attn_step_fwd')


def attn_pre(pm, w_prev, w_cum_prev, loc_conv_w, loc_lin_w, s_buf=None, parts=1):
    """S = pm + W_l conv([w_prev; w_cum_prev]): the part of the attention step that only needs the previous weights"""
    B, L, A = pm.shape
    F_, _, K = loc_conv_w.shape
    if s_buf is None:
        s_buf = torch.empty(B, L, A, device=pm.device, dtype=torch.float32)
    check(_lib.load().st_attn_pre_fwd(_p(pm), _p(w_prev), int(w_prev.stride(0)), _p(w_cum_prev), _p(loc_conv_w), _p(loc_lin_w),
                                      _p(s_buf), int(parts), B, L, A, F_, K, stream_handle()), 'st_attn_pre_fwd')
    return s_buf


def attn_fin(pq, s_buf, memory, w_cum_prev, v, w_out, w_cum_out, ctx, F_, K, parts=1):
    B, L, E = memory.shape
    A = s_buf.shape[-1]
    check(_lib.load().st_attn_fin_t16_fwd(_p(pq), _p(s_buf), _p(memory), _p(w_cum_prev), _p(w_out), int(w_out.stride(0)),
                                          _p(w_cum_out), _p(v), None, 0, _p(ctx), int(ctx.stride(0)), int(parts), B, L, A, E, int(F_), int(K),
                                          stream_handle()), 'st_attn_fin_t16_fwd')


def attn_fin_split(pq, s_buf, memory, w_cum_prev, v, w_out, w_cum_out, ctx, parts):
    """the fin part over `parts` position ranges + a combine launch (long texts): st_attn_fin_split_fwd"""
    lib = _lib.load()
    B, L, E = memory.shape
    A = s_buf.shape[-1]
    ws = torch.empty(int(lib.st_attn_fin_split_workspace_floats(B, E, int(parts))), device=memory.device, dtype=torch.float32)
    check(lib.st_attn_fin_split_fwd(_p(pq), _p(s_buf), _p(memory), _p(w_cum_prev), _p(w_out), int(w_out.stride(0)), _p(w_cum_out), _p(v),
                                    None, 0, _p(ctx), int(ctx.stride(0)), _p(ws), int(parts), B, L, A, E, stream_handle()),
          'st_attn_fin_split_fwd')


def query_attn_fin(packed_wq, h_q_t16, Q, s_buf, memory, w_cum_prev, v, w_out, w_cum_out, ctx_t16, F_, K, parts=2, epoch=1, granules=None):
    """query projection + attention fin part in ONE launch (pq handed over inside the launch): st_query_attn_fin_fwd.
    h_q_t16 / ctx_t16: T16 buffers of (B, Q) / (B, E); `granules` (B, 2 A floats) must be zero before the first epoch."""
    B, L, E = memory.shape
    A = s_buf.shape[-1]
    if granules is None:
        granules = torch.zeros(B, 2 * A, device=memory.device, dtype=torch.float32)
    job = _lib.StAttnFinJob()
    job.s_buf, job.memory, job.w_cum_prev = _p(s_buf), _p(memory), _p(w_cum_prev)
    job.w_out, job.ld_wout, job.w_cum_out, job.v = _p(w_out), int(w_out.stride(0)), _p(w_cum_out), _p(v)
    job.ctx_dst[0] = t16_view(ctx_t16, K=E)
    job.n_ctx_dst, job.parts, job.L, job.A, job.E, job.F, job.K = 1, int(parts), L, A, E, int(F_), int(K)
    hv = t16_view(h_q_t16, K=Q)
    check(_lib.load().st_query_attn_fin_fwd(_p(packed_wq), C.byref(hv), 16 * kb16(Q), _p(granules), int(epoch), C.byref(job), B,
                                            stream_handle()), 'st_query_attn_fin_fwd')
    return granules


def query_attn_rng(packed_wq, h_q_t16, Q, s_buf, memory, w_cum_prev, v, w_out, w_cum_out, ctx_t16, parts, epoch=1, granules=None,
                   xchg=None, status=None):
    """query projection + attention fin part over `parts` position ranges + combine in ONE launch: st_query_attn_rng_fwd.
    `granules` (B, 2 A floats) and `xchg` must be zero before the first epoch; returns them."""
    lib = _lib.load()
    B, L, E = memory.shape
    A = s_buf.shape[-1]
    if granules is None:
        granules = torch.zeros(B, 2 * A, device=memory.device, dtype=torch.float32)
    if xchg is None:
        xchg = torch.zeros(2 * int(lib.st_attn_rng_xchg_words(B, E, int(parts))), device=memory.device, dtype=torch.float32)
    job = _lib.StAttnFinJob()
    job.s_buf, job.memory, job.w_cum_prev = _p(s_buf), _p(memory), _p(w_cum_prev)
    job.w_out, job.ld_wout, job.w_cum_out, job.v = _p(w_out), int(w_out.stride(0)), _p(w_cum_out), _p(v)
    job.ctx_dst[0] = t16_view(ctx_t16, K=E)
    job.n_ctx_dst, job.parts, job.L, job.A, job.E, job.F, job.K = 1, int(parts), L, A, E, 0, 0
    job.status = _p(status, torch.int32)
    hv = t16_view(h_q_t16, K=Q)
    check(lib.st_query_attn_rng_fwd(_p(packed_wq), C.byref(hv), 16 * kb16(Q), _p(granules), _p(xchg), int(epoch), C.byref(job), B,
                                    stream_handle()), 'st_query_attn_rng_fwd')
    return granules, xchg


class _ParamLayouts:
    """Cached re-layouts of PARAMETERS (a layout change only: mode 0 = Conv1d weight (N, Cin, KT) -> tap-major (N, KT, Cin), the forward
    implicit-GEMM operand; mode 1 = (N, Cin, KT) -> (Cin, KT, N) with the taps reversed, the weight of the input-gradient conv, for
    KT = 1 the transpose of a Linear weight).  Keyed on (storage address, shape, mode); an entry holds a detached alias of the
    parameter (same storage and version counter) and is valid while the version counter has not moved.  When ANY entry is found
    stale (the optimiser has stepped), ALL entries are refreshed by ONE launch (st_relayout_batch): training re-lays out ~40 weights
    per step, one torch copy each in round 2.  Entries nobody asked for during the last 8 epochs (steps) are dropped -- except those a
    hipGraph capture has seen (the graph holds the buffer's address; a refresh rewrites it in place, so a replay after a weight
    update reads the new layout)."""
    MAX_ENTRIES = 1024

    def __init__(self):
        self.entries = {}       # key -> [alias, dst, version, (N, Cin, KT), epoch of the last use, pinned(, row stride of dst)]
        self.cats = {}          # key -> (dst, keys of its parts in `entries`): see get_cat
        self.table = None       # device descriptor table of the entries
        self.blk_desc = None    # descriptor index of every workgroup of the refresh launch
        self.count = 0
        self.total_blocks = 0
        self.dirty = True
        self.refreshes = 0      # launches
        self.epoch = 0          # stale-triggered refreshes (~ optimisation steps)

    def get(self, w, mode):
        key = (w.data_ptr(), tuple(w.shape), mode)
        e = self.entries.get(key)
        if e is not None:
            e[4] = self.epoch
            if capturing():
                e[5] = True                  # a hipGraph now holds this buffer's address: the entry is never dropped (and refreshed in place)
            if e[2] != e[0]._version:
                self.epoch += 1              # (a weight has moved: a new optimisation step -- entries age by these, not by launches)
                e[4] = self.epoch
                self.refresh()
            return e[1]
        N, Cin = int(w.shape[0]), int(w.shape[1])
        KT = int(w.shape[2]) if w.dim() == 3 else 1
        assert w.is_contiguous() and w.dtype == torch.float32
        if len(self.entries) >= self.MAX_ENTRIES:          # (inference-only processes never refresh: bound the table by age)
            for k in [k for k in sorted(self.entries, key=lambda k: self.entries[k][4]) if not self.entries[k][5]][:self.MAX_ENTRIES // 2]:
                del self.entries[k]
        dst = torch.empty((N, KT, Cin) if mode == 0 else (Cin, KT, N), device=w.device, dtype=torch.float32)
        self.entries[key] = [w.detach(), dst, -1, (N, Cin, KT), self.epoch, capturing()]
        self.dirty = True
        self.refresh()
        return dst

    def get_cat(self, ws, mode, pad_to=0):
        """Several parameters of equal Cin laid out side by side as ONE operand: mode 0 -> (sum N, Cin), their rows one after the other
        (the weight of one forward product in place of len(ws); 1-D parameters: their concatenation); mode 1 -> (Cin, sum N), their
        transposes as column blocks (the weight of one input-gradient product).  Same caching and the same single refresh launch as get()."""
        key = (tuple(w.data_ptr() for w in ws), tuple(tuple(w.shape) for w in ws), mode, 'cat', int(pad_to))
        c = self.cats.get(key)
        if c is not None and all(k in self.entries for k in c[1]):
            stale = False
            for k in c[1]:
                e = self.entries[k]
                e[4] = self.epoch
                if capturing():
                    e[5] = True
                stale = stale or e[2] != e[0]._version
            if stale:
                self.epoch += 1
                for k in c[1]:
                    self.entries[k][4] = self.epoch
                self.refresh()
            return c[0]
        one_d = ws[0].dim() == 1
        Ns = [int(w.shape[0]) for w in ws]
        Cin = 1 if one_d else int(ws[0].shape[1])
        assert all(w.is_contiguous() and w.dtype == torch.float32 and w.dim() == ws[0].dim() and (one_d or int(w.shape[1]) == Cin) for w in ws)
        Nt = sum(Ns)
        if pad_to:      # mode 1 only: (Cin, pad_to) with ZERO columns past sum N (a reduction axis padded to whole 16-byte pieces)
            assert mode == 1 and not one_d and pad_to >= Nt
            Nt_alloc = int(pad_to)
            dst = torch.zeros(Cin, Nt_alloc, device=ws[0].device, dtype=torch.float32)
        else:
            Nt_alloc = Nt
            dst = torch.empty((Nt,) if one_d else ((Nt, Cin) if mode == 0 else (Cin, Nt)), device=ws[0].device, dtype=torch.float32)
        keys, off = [], 0
        for w, N in zip(ws, Ns):
            part = dst[off:off + N] if (mode == 0 or one_d) else dst[:, off:off + N]
            k = (w.data_ptr(), tuple(w.shape), mode, 'part', dst.data_ptr(), off)
            self.entries[k] = [w.detach(), part, -1, (N, Cin, 1), self.epoch, capturing(), Nt_alloc if (mode == 1 and not one_d) else 0]
            keys.append(k)
            off += N
        self.cats[key] = (dst, keys)
        self.dirty = True
        self.refresh()
        return dst

    def refresh(self):
        lib = _lib.load()
        self.refreshes += 1
        drop = [k for k, e in self.entries.items() if self.epoch - e[4] > 8 and not e[5]]
        for k in drop:
            del self.entries[k]
            self.dirty = True
        if drop:
            self.cats = {k: c for k, c in self.cats.items() if all(pk in self.entries for pk in c[1])}
        if not self.entries:
            return
        if self.dirty:
            dev = next(iter(self.entries.values()))[1].device
            assert all(e[1].device == dev for e in self.entries.values()), 'parameter layouts of several devices in one process'
            arr = (_lib.StRelayoutDesc * len(self.entries))()
            blk, owner = 0, []
            for j, (d, (key, e)) in enumerate(zip(arr, self.entries.items())):
                N, Cin, KT = e[3]
                d.src, d.dst, d.N, d.Cin, d.KT, d.mode, d.blk0 = e[0].data_ptr(), e[1].data_ptr(), N, Cin, KT, key[2], blk
                d.ld_dst = e[6] if len(e) > 6 else 0
                nb = int(lib.st_relayout_blocks(N, Cin, KT))
                owner += [j] * nb
                blk += nb
            self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            self.blk_desc = torch.tensor(owner, dtype=torch.int32).to(dev)        # the descriptor of every workgroup
            self.count, self.total_blocks, self.dirty = len(self.entries), blk, False
        check(lib.st_relayout_batch(_p(self.table, torch.uint8), self.count, self.total_blocks, _p(self.blk_desc, torch.int32), stream_handle()),
              'st_relayout_batch')
        for e in self.entries.values():
            e[2] = e[0]._version


_LAYOUTS = _ParamLayouts()


def _tap_major(w):
    """Conv1d weight (N, Cin, KT) -> (N, KT, Cin) contiguous (layout change of a PARAMETER; a k-block of the implicit GEMM is then
    16-byte loadable).  Cached per tensor object and version: inference converts a weight once, training once per step -- and then
    all weights in one launch (_ParamLayouts)."""
    if w.is_contiguous() and w.is_cuda and w.dtype == torch.float32:
        return _LAYOUTS.get(w, 0)
    return w.detach().permute(0, 2, 1).contiguous()


def dx_weight(w):
    """the weight operand of the input-gradient product of a conv / linear layer with parameter w ((N, Cin, KT) or (N, Cin)):
    returns (weight, tap_major) for ops.gemm -- the cached (Cin, KT, N) tap-reversed form when the tap-major kernel can take it,
    otherwise torch's permute / flip copy in the Conv1d layout"""
    KT = w.shape[2] if w.dim() == 3 else 1
    N = w.shape[0]
    if w.is_contiguous() and w.is_cuda and w.dtype == torch.float32 and (KT == 1 or (N % 4 == 0 and N >= 16)):
        wt = _LAYOUTS.get(w, 1)
        return (wt.view(wt.shape[0], N), False) if KT == 1 else (wt, True)
    return (w.permute(1, 0, 2).flip(2) if w.dim() == 3 else w.t()).contiguous(), False


def gemm(a, w, out=None, *, Bn=None, Tin=None, Tout=None, pad=0, stride=1, coff=0, bias=None, act_pre=None,
         bn=None, bn_eps=1e-5, act_post=None, res=None, highway_h=None, mask=None, pool_prev=False, w_tap_major=False,
         collect=None):
    """C = epilogue(conv1d / linear).  a: (Bn, Tin, Cin) or (M, Cin) channels-last; w: torch Linear
    (N, Cin) or Conv1d (N, Cin, KT) weight -- or, with w_tap_major, a conv weight already in the tap-major (N, KT, Cin) layout
    (ops.dx_weight).  bn = (mean, var, weight, bias) tensors.  collect: a list -- the job is appended to it instead of being
    launched (ops.gemm_flush(list) then runs all collected jobs, sharing launches where the kernels allow)."""
    lib = _lib.load()
    if a.dim() == 3:
        Bn_, Tin_, Cin = a.shape
    else:
        Bn_, Tin_, Cin = 1, a.shape[0], a.shape[1]
    Bn = Bn or Bn_
    Tin = Tin or Tin_
    KT = (w.shape[1] if w_tap_major else w.shape[2]) if w.dim() == 3 else 1
    N = w.shape[0]
    if w_tap_major:
        assert w.dim() == 3 and w.shape[2] == Cin and w.is_contiguous() and KT > 1 and Cin % 4 == 0
    if Tout is None:
        Tout = (Tin + 2 * pad - KT) // stride + 1
    lda = a.stride(-2)
    if out is None:
        shape = (Bn, Tout, N) if a.dim() == 3 else (Bn * Tout, N)
        out = torch.empty(shape, device=a.device, dtype=torch.float32)
    ldc = out.stride(-2)
    ep = StGemmEpilogue()
    ep.bias = _p(bias)
    ep.act_pre = ACT[act_pre]
    if bn is not None:
        ep.bn_mean, ep.bn_var, ep.bn_w, ep.bn_b = _p(bn[0]), _p(bn[1]), _p(bn[2]), _p(bn[3])
    ep.bn_eps = float(bn_eps)
    ep.act_post = ACT[act_post]
    ep.res = _p(res)
    ep.ldres = int(res.stride(-2)) if res is not None else 0
    ep.highway_h = _p(highway_h)
    ep.ldhw = int(highway_h.stride(-2)) if highway_h is not None else 0
    ep.mask = _p(mask)
    ep.ldmask = int(mask.stride(-2)) if mask is not None else 0
    if w_tap_major:
        ep.w_tap_major = 1
    elif KT > 1 and Cin % 4 == 0 and Cin >= 16:
        w = _tap_major(w)
        ep.w_tap_major = 1
    if pool_prev and ep.w_tap_major and a.dim() == 3 and a.is_contiguous() and stride == 1 and a.data_ptr() % 16 == 0 and \
            (Bn, Tin) == (a.shape[0], a.shape[1]):
        # the LDS-DMA kernel cannot take a maximum on the way into LDS: the pooled input becomes a tensor of its own (one elementwise
        # launch, ~10 us for the CBHG's (32, 258, 640) bank output) and the conv runs on the fast kernel (52 against 84 us)
        a, pool_prev = pool_prev_fwd(a), False
    slabs = int(lib.st_gemm_splitk_slabs(int(Bn), int(Tout), int(Cin), int(N), int(KT)))
    ws = None
    if slabs > 1:         # small grid, long reduction: partial products per k range + a finish pass (st_gemm_epilogue.splitk_ws)
        ws = torch.empty(slabs * Bn * Tout * N, device=a.device, dtype=torch.float32)
        ep.splitk_ws, ep.splitk_slabs = _p(ws), slabs
    if collect is not None:
        job = _lib.StGemmJob()
        job.A, job.lda, job.W, job.C, job.ldc, job.coff = _p(a), int(lda), _p(w), _p(out), int(ldc), int(coff)
        job.Bn, job.Tin, job.Tout, job.Cin, job.N, job.KT = int(Bn), int(Tin), int(Tout), int(Cin), int(N), int(KT)
        job.pad, job.stride, job.pool_prev, job.ep = int(pad), int(stride), 1 if pool_prev else 0, ep
        collect.append((job, (a, w, out, ws, bias, bn, res, highway_h, mask)))       # (the tensors stay alive until the flush)
        return out
    check(lib.st_gemm_fwd(_p(a), int(lda), _p(w), _p(out), int(ldc), int(coff), int(Bn), int(Tin), int(Tout),
                          int(Cin), int(N), int(KT), int(pad), int(stride), 1 if pool_prev else 0, C.byref(ep), stream_handle()),
          'st_gemm_fwd')
    return out


def highway_stack(x, layers):
    """eval-mode highway stack in one launch; layers: [(W_H, b_H, W_T, b_T), ...] torch Linear parameters.  None when the shape is
    not one the kernel takes (the caller then runs the layers as GEMM pairs)."""
    lib = _lib.load()
    Cn = int(x.shape[-1])
    if not (x.is_contiguous() and lib.st_highway_stack_supported(Cn, len(layers))) or any(
            tuple(w.shape) != (Cn, Cn) or not w.is_contiguous() for l in layers for w in (l[0], l[2])):
        return None
    n = len(layers)
    arr = lambda k: (C.c_void_p * n)(*[_p(l[k]) for l in layers])
    y = torch.empty_like(x)
    M = x.numel() // Cn
    check(lib.st_highway_stack_fwd(_p(x), Cn, arr(0), arr(1), arr(2), arr(3), n, _p(y), Cn, int(M), Cn, stream_handle()),
          'st_highway_stack_fwd')
    return y


def gemm_flush(collected):
    """run the jobs gathered by ops.gemm(..., collect=list) -- st_gemm_fwd_batch: one launch for up to eight jobs of the pipelined
    kernel (the conv bank), separate launches otherwise"""
    if not collected:
        return
    arr = (_lib.StGemmJob * len(collected))(*[j for j, _ in collected])
    check(_lib.load().st_gemm_fwd_batch(arr, len(collected), stream_handle()), 'st_gemm_fwd_batch')
    del collected[:]


def _colreduce_ws(M, N, device):
    n = int(_lib.load().st_colreduce_workspace_floats(int(M), int(N)))
    return torch.empty(n, device=device, dtype=torch.float32)


def bn_stats(x2d, coff, N, run_mean=None, run_var=None, momentum=0.1, batches_tracked=None):
    """per-column batch statistics of x2d[:, coff:coff+N] (+ running-stat update in place, + `num_batches_tracked` += 1)"""
    lib = _lib.load()
    M = x2d.shape[0]
    mean = torch.empty(N, device=x2d.device, dtype=torch.float32)
    var = torch.empty(N, device=x2d.device, dtype=torch.float32)
    check(lib.st_bn_stats(_p(x2d), int(x2d.stride(0)), int(coff), M, N, _p(mean), _p(var), _p(run_mean), _p(run_var),
                          float(momentum), _p(batches_tracked, torch.int64), _p(_colreduce_ws(M, N, x2d.device)), stream_handle()),
          'st_bn_stats')
    return mean, var


def bn_stats_record(x2d, coff, N, batches_tracked=None):
    """SyncBN, local half: (mean[N], M2[N], row count) of this rank's rows of x2d[:, coff:coff+N], as one (2N + 1) record"""
    M = x2d.shape[0]
    rec = torch.empty(2 * N + 1, device=x2d.device, dtype=torch.float32)
    check(_lib.load().st_bn_stats_record(_p(x2d), int(x2d.stride(0)), int(coff), M, N, _p(rec), _p(batches_tracked, torch.int64),
                                         _p(_colreduce_ws(M, N, x2d.device)), stream_handle()), 'st_bn_stats_record')
    return rec


def bn_sync_merge(rec, N, run_mean=None, run_var=None, momentum=0.1):
    """SyncBN, merged half: the gathered (world, 2N + 1) records -> (mean, var, 1 / global row count) of the global batch in one
    launch (+ running-stat update in place)"""
    world = rec.shape[0]
    assert rec.is_contiguous() and rec.shape[1] == 2 * N + 1
    out = torch.empty(2 * N + 1, device=rec.device, dtype=torch.float32)
    mean, var, inv_total = out[:N], out[N:2 * N], out[2 * N:]
    check(_lib.load().st_bn_sync_merge(_p(rec), world, N, _p(mean), _p(var), _p(run_mean), _p(run_var), float(momentum), _p(inv_total),
                                       stream_handle()), 'st_bn_sync_merge')
    return mean, var, inv_total


def layer_norm(x2d, gamma, beta, eps, want_stats=False):
    """nn.LayerNorm over the last dimension of (M, N) rows -> y [, mean (M), rstd (M)]"""
    M, N = x2d.shape
    y = torch.empty(M, N, device=x2d.device, dtype=torch.float32)
    mean = torch.empty(M, device=x2d.device, dtype=torch.float32) if want_stats else None
    rstd = torch.empty(M, device=x2d.device, dtype=torch.float32) if want_stats else None
    check(_lib.load().st_layer_norm_fwd(_p(x2d), int(x2d.stride(0)), _p(gamma), _p(beta), float(eps), _p(y), N, _p(mean), _p(rstd), M, N,
                                        stream_handle()), 'st_layer_norm_fwd')
    return (y, mean, rstd) if want_stats else y


def layer_norm_bwd(dy2d, x2d, gamma, mean, rstd):
    """-> (dx, dy * xhat): d gamma = colsum(dy * xhat), d beta = colsum(dy)"""
    M, N = x2d.shape
    dx = torch.empty(M, N, device=x2d.device, dtype=torch.float32)
    dyxhat = torch.empty(M, N, device=x2d.device, dtype=torch.float32)
    check(_lib.load().st_layer_norm_bwd(_p(dy2d), int(dy2d.stride(0)), _p(x2d), int(x2d.stride(0)), _p(gamma), _p(mean), _p(rstd),
                                        _p(dx), N, _p(dyxhat), M, N, stream_handle()), 'st_layer_norm_bwd')
    return dx, dyxhat


def log_softmax(x):
    N = x.shape[-1]
    x = x.contiguous()
    y = torch.empty_like(x)
    check(_lib.load().st_log_softmax_fwd(_p(x), _p(y), x.numel() // N, N, stream_handle()), 'st_log_softmax_fwd')
    return y


def log_softmax_bwd(dy, y):
    N = y.shape[-1]
    dx = torch.empty_like(y)
    check(_lib.load().st_log_softmax_bwd(_p(dy.contiguous()), _p(y), _p(dx), y.numel() // N, N, stream_handle()), 'st_log_softmax_bwd')
    return dx


def bn_apply(x2d, coff, N, mean, var, w, b, eps, act=None):
    lib = _lib.load()
    check(lib.st_bn_apply(_p(x2d), int(x2d.stride(0)), int(coff), x2d.shape[0], N, _p(mean), _p(var), _p(w), _p(b),
                          float(eps), ACT[act], stream_handle()), 'st_bn_apply')


def lstm_seq(xproj, w_hh, b_hh, out, ocol, reverse, ws=None, gates_tape=None, c_tape=None):
    lib = _lib.load()
    B, T, H4 = xproj.shape
    H = H4 // 4
    if ws is None:
        ws = torch.empty(3 * B * H, device=xproj.device, dtype=torch.float32)
    check(lib.st_lstm_seq_fwd(_p(xproj), _p(w_hh), _p(b_hh), _p(out), int(out.stride(1)), int(ocol), _p(ws),
                              _p(gates_tape), _p(c_tape), B, T, H, 1 if reverse else 0, stream_handle()), 'st_lstm_seq_fwd')


LSTM_PERSIST = True      # bidirectional LSTM layers as one launch for all steps where the shape allows (st_lstm_seq2_persist_fwd)
_PERSIST_STATUS = {}


def _dev_index(device):
    device = torch.device(device)
    return device.index if device.index is not None else torch.cuda.current_device()


def persist_status(device):
    """the device word the one-launch recurrent layers report a starved launch in (bit 1); see check_persist_status"""
    key = _dev_index(device)
    t = _PERSIST_STATUS.get(key)
    if t is None:
        t = _PERSIST_STATUS[key] = torch.zeros(1, device=torch.device('cuda', key), dtype=torch.int32)
    return t


def persist_starved(device=None):
    """True (and the word cleared) if a one-launch recurrent layer gave up waiting for its neighbour workgroups since the last check.
    Reads one device word per device (synchronises)."""
    hit = False
    for key, t in list(_PERSIST_STATUS.items()):
        if device is not None and key != _dev_index(device):
            continue
        if int(t.item()):
            t.zero_()
            hit = True
    return hit


def check_persist_status(device=None):
    """Raise if a one-launch recurrent layer gave up waiting for its neighbour workgroups since the last check (the launch was starved
    of compute units; the rows it produced are NaN).  Reads one device word per device (synchronises) and clears it."""
    for key, t in list(_PERSIST_STATUS.items()):
        if device is not None and key != _dev_index(device):
            continue
        v = int(t.item())
        if v:
            t.zero_()
            raise Starved('one-launch LSTM layer: a wait for the hidden state timed out (status word 0x%x): the launch was starved '
                               'of compute units and its outputs are NaN.  Set semi_tts_amd.ops.LSTM_PERSIST = False to run the layer '
                               'as one launch per time step.' % v)


def lstm_seq2(xproj_f, xproj_b, w_hh_f, w_hh_b, b_hh_f, b_hh_b, out, gates_tapes=None, c_tapes=None):
    """both directions of a bidirectional LSTM layer; out (B,T,2H): forward direction in columns [0,H), reverse in [H,2H).
    One launch for the whole layer where st_lstm_seq2_persist_supported takes the shape, else one launch per time step."""
    lib = _lib.load()
    B, T, H4 = xproj_f.shape
    H = H4 // 4
    P2 = C.c_void_p * 2
    arr = lambda a, b_: P2(_p(a), _p(b_))
    g2 = arr(*gates_tapes) if gates_tapes is not None else None
    c2 = arr(*c_tapes) if c_tapes is not None else None
    if LSTM_PERSIST and out.stride(2) == 1 and out.stride(0) == T * out.stride(1) and \
            all(t.data_ptr() % 16 == 0 for t in (out, w_hh_f, w_hh_b)) and w_hh_f.is_contiguous() and w_hh_b.is_contiguous() and \
            lib.st_lstm_seq2_persist_supported(B, T, H, int(out.stride(1)), 0, H):
        join_side()               # (its workgroups wait for each other inside the launch: nothing else may hold compute units meanwhile)
        # all T steps in one launch (recurrent weights in registers, h handed over through `out` itself)
        check(lib.st_lstm_seq2_persist_fwd(arr(xproj_f, xproj_b), arr(w_hh_f, w_hh_b), arr(b_hh_f, b_hh_b), _p(out), int(out.stride(1)),
                                           (C.c_int * 2)(0, H), g2, c2, B, T, H, _p(persist_status(out.device), torch.int32), stream_handle()),
              'st_lstm_seq2_persist_fwd')
        return
    ws = torch.empty(6 * B * H, device=xproj_f.device, dtype=torch.float32)
    check(lib.st_lstm_seq2_fwd(arr(xproj_f, xproj_b), arr(w_hh_f, w_hh_b), arr(b_hh_f, b_hh_b), _p(out), int(out.stride(1)),
                               (C.c_int * 2)(0, H), _p(ws), g2, c2, B, T, H, stream_handle()), 'st_lstm_seq2_fwd')


def lstm_seq_bwd(dout, dcol, gates_tape, c_tape, w_hh_t, reverse):
    """dout (B,T,>=dcol+H) -> dxproj (B,T,4H)"""
    T, B, _, H = gates_tape.shape
    dxproj = torch.empty(B, T, 4 * H, device=dout.device, dtype=torch.float32)
    ws = torch.empty(2 * B * H, device=dout.device, dtype=torch.float32)
    check(_lib.load().st_lstm_seq_bwd(_p(dout), int(dout.stride(1)), int(dcol), _p(gates_tape), _p(c_tape), _p(w_hh_t),
                                      _p(dxproj), _p(ws), B, T, H, 1 if reverse else 0, stream_handle()), 'st_lstm_seq_bwd')
    return dxproj


PERSIST_CU_RESERVE = int(os.environ.get('ST_PERSIST_CU_RESERVE', '32'))      # compute units left to RCCL's channels beside the one-launch BiLSTM backward
_N_CU = {}


def _persist_bwd_headroom(B, H):
    """The one-launch BiLSTM backward needs all of its workgroups co-resident.  Under data parallelism the decoder's gradient buckets are
    all-reduced by RCCL kernels on another stream exactly while the encoder's BiLSTM backward runs (advisor, round 5): with more than one
    rank and asynchronous buckets in flight the launch must leave PERSIST_CU_RESERVE compute units free, or the per-step form runs instead
    (a starved launch costs a skipped step and switches LSTM_PERSIST off for good)."""
    sink = _GRAD_SINK
    if sink is None or not getattr(sink, '_active', False):
        return True
    try:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return True
    except Exception:
        return True
    if not any(w is not None for w in getattr(sink, 'works', ())):
        return True
    dev = torch.cuda.current_device()
    if dev not in _N_CU:
        _N_CU[dev] = device_info()['n_cu']
    return 2 * (H // 16) * ((B + 15) // 16) <= _N_CU[dev] - PERSIST_CU_RESERVE


def lstm_seq2_bwd(dout, gates_tapes, c_tapes, w_hh_ts, w_hhs=None):
    """both directions at once: dout (B,T,2H) -> (dxproj_f, dxproj_b), each (B,T,4H).  w_hhs = the two W_hh parameters (4H, H): with
    H % 16 == 0 the loop runs on packed operands, one launch per step (product + pointwise backward of the previous step)."""
    T, B, _, H = gates_tapes[0].shape
    dx = list(torch.empty(2, B, T, 4 * H, device=dout.device, dtype=torch.float32).unbind(0))      # (back to back: one sentinel fill)
    lib = _lib.load()
    P2 = C.c_void_p * 2
    arr = lambda a, b_: P2(_p(a), _p(b_))
    if LSTM_PERSIST and w_hhs is not None and dout.stride(2) == 1 and dout.stride(0) == T * dout.stride(1) and dout.data_ptr() % 16 == 0 and \
            all(w.is_contiguous() for w in w_hhs) and lib.st_lstm_seq2_bwd_persist_supported(B, T, H, int(dout.stride(1)), 0, H) and \
            _persist_bwd_headroom(B, H):
        # all T steps in one launch (slices of W_hh and the carried dL/dc in registers, the gate gradients handed over through dx itself)
        join_side()
        check(lib.st_lstm_seq2_bwd_persist(_p(dout), int(dout.stride(1)), (C.c_int * 2)(0, H), arr(*gates_tapes), arr(*c_tapes),
                                           arr(*[w.detach() for w in w_hhs]), arr(*dx), B, T, H,
                                           _p(persist_status(dout.device), torch.int32), stream_handle()), 'st_lstm_seq2_bwd_persist')
        return dx
    if w_hhs is not None and H % 16 == 0:
        packed = [pack_weight_t([w.detach()]) for w in w_hhs]                 # W_hh^T in MFMA order: N = H, K = 4H
        ws = torch.empty(2 * B * H, device=dout.device, dtype=torch.float32)
        t16 = torch.empty(4 * t16_floats(B, 4 * H), device=dout.device, dtype=torch.float32)
        check(_lib.load().st_lstm_seq2_bwd_packed(_p(dout), int(dout.stride(1)), (C.c_int * 2)(0, H), arr(*gates_tapes), arr(*c_tapes),
                                                  arr(*packed), arr(*dx), _p(ws), _p(t16), B, T, H, stream_handle()),
              'st_lstm_seq2_bwd_packed')
        return dx
    ws = torch.empty(4 * B * H, device=dout.device, dtype=torch.float32)
    P2 = C.c_void_p * 2
    arr = lambda a, b_: P2(_p(a), _p(b_))
    check(_lib.load().st_lstm_seq2_bwd(_p(dout), int(dout.stride(1)), (C.c_int * 2)(0, H), arr(*gates_tapes), arr(*c_tapes),
                                       arr(*w_hh_ts), arr(*dx), _p(ws), B, T, H, stream_handle()), 'st_lstm_seq2_bwd')
    return dx


def gru_seq(gi_f, gi_b, w_hh_f, w_hh_b, b_hh_f, b_hh_b, out, tape=None):
    lib = _lib.load()
    B, T, H3 = gi_f.shape
    ndir = 2 if gi_b is not None else 1
    check(lib.st_gru_seq_fwd(_p(gi_f), _p(gi_b), _p(w_hh_f), _p(w_hh_b), _p(b_hh_f), _p(b_hh_b), _p(out),
                             int(out.stride(1)), _p(tape), B, T, H3 // 3, ndir, stream_handle()), 'st_gru_seq_fwd')


def gru_seq_bwd(dout, out, tape, w_hh_f, w_hh_b):
    """returns (dgi_f, dgi_b, dgh_f, dgh_b), each (B,T,3H)"""
    ndir, B, T, _, H = tape.shape
    mk = lambda: torch.empty(B, T, 3 * H, device=dout.device, dtype=torch.float32)
    dgi_f, dgh_f = mk(), mk()
    dgi_b, dgh_b = (mk(), mk()) if ndir == 2 else (None, None)
    check(_lib.load().st_gru_seq_bwd(_p(dout), int(dout.stride(1)), _p(out), int(out.stride(1)), _p(tape), _p(w_hh_f),
                                     _p(w_hh_b), _p(dgi_f), _p(dgi_b), _p(dgh_f), _p(dgh_b), B, T, H, ndir,
                                     stream_handle()), 'st_gru_seq_bwd')
    return dgi_f, dgi_b, dgh_f, dgh_b


def vq_build_table(learnable, attr=None, attr_w=None, attr_b=None):
    lib = _lib.load()
    V, Dl = learnable.shape
    Da = attr_w.shape[0] if attr_w is not None else 0
    table = torch.empty(V, Dl + Da, device=learnable.device, dtype=torch.float32)
    check(lib.st_vq_build_table(_p(learnable), Dl, _p(attr), attr.shape[1] if attr is not None else 0, _p(attr_w),
                                _p(attr_b), Da, _p(table), V, stream_handle()), 'st_vq_build_table')
    return table


def gather_rows(table, idx):
    lib = _lib.load()
    V, D = table.shape
    idx = idx.contiguous()
    out = torch.empty(tuple(idx.shape) + (D,), device=table.device, dtype=torch.float32)
    check(lib.st_gather_rows(_p(table), _p(idx, torch.int64), _p(out), idx.numel(), D, V, stream_handle()), 'st_gather_rows')
    return out


def vq_mfma_shape(D, V):
    """shapes the matrix-core nearest-code search takes (st_vq_pack_table / st_vq_l2_packed_fwd)"""
    return D <= 64 and D % 4 == 0 and V <= 1024


def vq_pack_table(table):
    """MFMA-order copy of a (V, D) code table: made once per table version (embed.L2Embedding caches it), not per lookup"""
    lib = _lib.load()
    V, D = table.shape
    packed = torch.empty(int(lib.st_vq_l2_workspace_floats(D, V)), device=table.device, dtype=torch.float32)
    check(lib.st_vq_pack_table(_p(table), _p(packed), D, V, stream_handle()), 'st_vq_pack_table')
    return packed


def vq_l2(x, table, temp, scalar_kernel=False, packed=None):
    """scalar_kernel=True (tests): the LDS-table kernel instead of the matrix-core one; packed: vq_pack_table(table)"""
    lib = _lib.load()
    lead, D = x.shape[:-1], x.shape[-1]
    V = table.shape[0]
    n = x.numel() // D
    p = torch.empty(tuple(lead) + (V,), device=x.device, dtype=torch.float32)
    idx = torch.empty(tuple(lead), device=x.device, dtype=torch.int64)
    out = torch.empty_like(x)
    if packed is not None and not scalar_kernel:
        check(lib.st_vq_l2_packed_fwd(_p(x), _p(table), _p(packed), _p(temp), _p(p), _p(idx, torch.int64), _p(out), n, D, V,
                                      stream_handle()), 'st_vq_l2_packed_fwd')
        return p, idx, out
    ws = None if scalar_kernel else torch.empty(int(lib.st_vq_l2_workspace_floats(D, V)), device=x.device, dtype=torch.float32)
    check(lib.st_vq_l2_fwd(_p(x), _p(table), _p(temp), _p(p), _p(idx, torch.int64), _p(out), _p(ws), n, D, V, stream_handle()),
          'st_vq_l2_fwd')
    return p, idx, out


def softmax_bwd(p, dp, scale=1.0, relu_scale=None, want_rowsum=False):
    """dz = s * p * (dp - sum(dp * p, -1)) with s = scale * relu(relu_scale[0]); optionally also dz.sum(-1)"""
    V = p.shape[-1]
    n = p.numel() // V
    dz = torch.empty_like(p)
    rs = torch.empty(p.shape[:-1], device=p.device, dtype=torch.float32) if want_rowsum else None
    check(_lib.load().st_softmax_bwd(_p(p), _p(dp), _p(relu_scale), float(scale), _p(dz), _p(rs), n, V, stream_handle()), 'st_softmax_bwd')
    return (dz, rs) if want_rowsum else dz


def rowscale_combine(a, alpha, x=None, r=None, beta=0.0, c=None):
    """alpha * a + beta * r[:, None] * x (+ c), all (M, D) row-major"""
    M, D = a.shape
    out = torch.empty_like(a)
    check(_lib.load().st_rowscale_combine(_p(a), float(alpha), _p(x), _p(r), float(beta), _p(c), _p(out), M, D, stream_handle()),
          'st_rowscale_combine')
    return out


def ctc_loss(prob, text, eps, want_grad=True, log_input=False):
    """-> (loss scalar tensor, d loss / d prob or None); see st_ctc_loss (log_input: prob holds log-probabilities)"""
    lib = _lib.load()
    B, T, V = prob.shape
    L = text.shape[1]
    loss = torch.empty((), device=prob.device, dtype=torch.float32)
    dprob = torch.empty_like(prob) if want_grad else None
    ws = torch.empty(int(lib.st_ctc_workspace_floats(B, T)), device=prob.device, dtype=torch.float32)
    check(lib.st_ctc_loss(_p(prob), _p(text, torch.int64), float(eps), _p(loss), _p(dprob), _p(ws), B, T, V, L, 1 if log_input else 0,
                          stream_handle()), 'st_ctc_loss')
    return loss, dprob


def softmax_argmax(logits):
    lib = _lib.load()
    V = logits.shape[-1]
    n = logits.numel() // V
    p = torch.empty_like(logits)
    idx = torch.empty(logits.shape[:-1], device=logits.device, dtype=torch.int64)
    check(lib.st_softmax_argmax(_p(logits), _p(p), _p(idx, torch.int64), n, V, stream_handle()), 'st_softmax_argmax')
    return p, idx


def fill_(t, v):
    check(_lib.load().st_fill(_p(t), float(v), t.numel(), stream_handle()), 'st_fill')
    return t


def copy2d(dst, src, rows, cols, ldd=None, lds=None):
    check(_lib.load().st_copy2d(_p(dst), int(ldd if ldd is not None else dst.stride(0)), _p(src),
                                int(lds if lds is not None else src.stride(0)), rows, cols, stream_handle()), 'st_copy2d')


def mean_rows(src):
    B, T, D = src.shape
    dst = torch.empty(B, D, device=src.device, dtype=torch.float32)
    check(_lib.load().st_mean_rows(_p(src), _p(dst), B, T, D, stream_handle()), 'st_mean_rows')
    return dst


# --------------------------------------------------------------------------------------------- backward blocks
def bn_norm(x2d, xoff, N, mean, var, w, b, eps, act=None, out=None, yoff=0):
    """out-of-place BatchNorm normalisation (training forward keeps x2d for the backward)"""
    M = x2d.shape[0]
    if out is None:
        out = torch.empty(M, N, device=x2d.device, dtype=torch.float32)
    check(_lib.load().st_bn_norm_fwd(_p(x2d), int(x2d.stride(0)), int(xoff), _p(out), int(out.stride(0)), int(yoff), M, N,
                                     _p(mean), _p(var), _p(w), _p(b), float(eps), ACT[act], stream_handle()), 'st_bn_norm_fwd')
    return out


def bn_norm_res_mask(x2d, mean, var, w, b, eps, act=None, res2d=None, mask2d=None, want_t=True):
    """-> (t, y): t = act(BatchNorm(x)) kept for the backward (None unless want_t), y = (t + res) * mask -- one launch (ConvLayer's tail,
    src/module.py:641-646).  Contiguous (M, N) operands."""
    M, N = x2d.shape
    y = torch.empty(M, N, device=x2d.device, dtype=torch.float32)
    t = torch.empty(M, N, device=x2d.device, dtype=torch.float32) if want_t and (res2d is not None or mask2d is not None) else None
    check(_lib.load().st_bn_norm_res_mask_fwd(_p(x2d), _p(t), _p(y), M, N, _p(mean), _p(var), _p(w), _p(b), float(eps), ACT[act],
                                              _p(res2d), _p(mask2d), stream_handle()), 'st_bn_norm_res_mask_fwd')
    return (t if t is not None else (y if want_t else None)), y


# ------------------------------------------------------------------------- where parameter gradients are written
_GRAD_SINK = None


def set_grad_sink(sink, only_if=None):
    """`sink.claim(param)` -> tensor or None decides where the kernels that produce parameter gradients write them: under data
    parallelism parallel.GradReducer hands out the parameter's slot in its all-reduce bucket, so the gradient is born in the
    communication buffer (no accumulate launch, no copy).  only_if: clear only when that sink is the registered one."""
    global _GRAD_SINK
    if only_if is not None and _GRAD_SINK is not only_if:
        return
    _GRAD_SINK = sink


def grad_slot(param):
    """the tensor the gradient of `param` should be written to (same shape), or None: allocate as usual"""
    s = _GRAD_SINK
    if s is None or param is None:
        return None
    return s.claim(param)


def mt_copy(dsts, srcs):
    """dsts[i] <- srcs[i] (contiguous fp32 device tensors of equal sizes) in a handful of multi-tensor launches"""
    import ctypes as C
    n = len(dsts)
    dp = (C.c_void_p * n)(*[t.data_ptr() for t in dsts])
    sp = (C.c_void_p * n)(*[t.data_ptr() for t in srcs])
    sz = (C.c_long * n)(*[t.numel() for t in dsts])
    check(_lib.load().st_mt_copy(dp, sp, sz, n, stream_handle()), 'st_mt_copy')


def gemm_wgrad_split(dc, a, split, w0=None, w1=None, b0=None, b1=None, with_db=False):
    """gemm_wgrad of a Linear over concatenated inputs, cut at input column `split`: (dW0 (N, split), dW1 (N, Cin - split)[, db, db'])
    -- the gradients of [W_ih | W_hh] (and b_ih, b_hh: equal) of an LSTM cell leave the product's fixed-order slab sum as the
    parameters' own gradient tensors (w0 / w1 / b0 / b1: the parameters, asked for their grad_slot)"""
    lib = _lib.load()
    M, Cin = a.shape
    N = dc.shape[-1]
    f32 = dict(device=a.device, dtype=torch.float32)
    d0, d1 = grad_slot(w0), grad_slot(w1)
    d0 = d0 if d0 is not None else torch.empty(N, split, **f32)
    d1 = d1 if d1 is not None else torch.empty(N, Cin - split, **f32)
    db = dbd = None
    if with_db:
        db, dbd = grad_slot(b0), grad_slot(b1)
        db = db if db is not None else torch.empty(N, **f32)
        dbd = dbd if dbd is not None else torch.empty(N, **f32)
    nws = int(lib.st_gemm_wgrad_workspace_floats(1, int(M), int(Cin), int(N), 1))
    ws = torch.empty(nws, **f32)
    check(lib.st_gemm_wgrad_split(_p(dc), int(dc.stride(-2)), 0, _p(a), int(a.stride(-2)), _p(d0), int(split), _p(d1),
                                  _p(db), _p(dbd), _p(ws), int(M), int(Cin), int(N), 0, stream_handle()), 'st_gemm_wgrad_split')
    return (d0, d1, db, dbd) if with_db else (d0, d1)


# Weight gradients are leaves of the backward pass: nothing later in it reads them.  Inside an autograd backward the products marked
# `later=True` are therefore only QUEUED (their output tensors exist at once, and are what the Function returns); the queue leaves as
# batched launches (st_gemm_wgrad_batch: one product launch + one slab-sum launch per 16 small products instead of two launches each,
# and small products share the chip) when it is full, before a gradient bucket is handed to the all-reduce (parallel.GradReducer),
# and from an engine callback at the end of the backward pass -- before `backward()` returns.  ST_WGRAD_DEFER=0: every product at once.
# A queued product must be the ONLY thing that has touched its parameter's gradient when it finally runs (it overwrites): the sites ask
# `grad_first(*params)` -- True when this is the first gradient contribution to each of them in this backward pass -- and queue only
# then; a later contribution to a parameter whose first one is still queued (the prenet's weights under partial teacher forcing: the
# decoder's own-output path and the teacher path) sends the queue out first and runs at once, so autograd's in-place accumulate finds
# both written.
_WQ = []
_WQ_TASK = -1
_WQ_PENDING = set()
_G_TASK = -1
_G_COUNT = {}
WGRAD_DEFER = os.environ.get('ST_WGRAD_DEFER', '1') != '0'
WGRAD_QMAX = 16


def _graph_task():
    f = getattr(torch._C, '_current_graph_task_id', None)
    return f() if f is not None else -1


def flush_wgrads():
    """launch the queued weight-gradient products (a no-op when there are none)"""
    global _WQ
    if _WQ:
        jobs, _WQ = _WQ, []
        _WQ_PENDING.clear()
        cur = stream_handle()
        streams = []
        for j in jobs:
            if j.get('_stream', cur) not in streams:
                streams.append(j.get('_stream', cur))
        for h in streams:
            group = [j for j in jobs if j.get('_stream', cur) == h]
            if h == cur:
                gemm_wgrad_batch(group)
            else:
                with use_stream(h):
                    gemm_wgrad_batch(group)


def grad_first(*params):
    """announce a gradient contribution to each of `params` (None entries ignored) by the calling backward function; True when it is
    the first one to every one of them in this backward pass and all of them are leaves (the returned gradient goes straight to
    `.grad`, nothing reads it on the way) -- the condition for handing the product to the queue (`later=`).  Outside a backward
    pass: False."""
    global _G_TASK
    tid = _graph_task() if WGRAD_DEFER else -1
    if tid == -1:
        return False
    if tid != _G_TASK:
        _G_TASK = tid
        _G_COUNT.clear()
    first = True
    for q in params:
        if q is None:
            continue
        k = q.data_ptr()
        n = _G_COUNT.get(k, 0) + 1
        _G_COUNT[k] = n
        if n > 1:
            first = False
            if k in _WQ_PENDING:
                flush_wgrads()
        if not q.is_leaf:                 # (computed from parameters by torch ops: their backward reads the gradient at once)
            first = False
        # a parameter that already HAS a gradient (a second backward() without zero_grad -- gradient accumulation --, or
        # zero_grad(set_to_none=False)) makes AccumulateGrad run `grad += new` the moment the function returns, and a tensor hook reads
        # the gradient there as well: both before a queued product would have written it.  Such a product is launched at once.
        elif q.grad is not None or q._backward_hooks:
            first = False
    return first


def _queue_wgrads(jobs, params=()):
    """jobs (dicts of gemm_wgrad_batch, outputs allocated) onto the queue; False outside a backward pass (the caller launches them)"""
    global _WQ_TASK
    tid = _graph_task() if WGRAD_DEFER else -1
    if tid == -1:
        return False
    if _WQ and _WQ_TASK != tid:
        _WQ.clear()                       # (left by a backward pass that raised: their tensors belong to a step that is gone)
        _WQ_PENDING.clear()
    if not _WQ:
        _WQ_TASK = tid
        torch.autograd.Variable._execution_engine.queue_callback(flush_wgrads)
    h = stream_handle()
    for j in jobs:
        j['_stream'] = h                  # (launched later, perhaps by another thread: on the stream its operands were produced on)
    _WQ.extend(jobs)
    _WQ_PENDING.update(q.data_ptr() for q in params if q is not None)
    if len(_WQ) >= WGRAD_QMAX:
        flush_wgrads()
    return True


def _alias(t):
    """a second tensor object over the same memory: autograd adopts a returned gradient as `.grad` only when nobody else holds that
    OBJECT (otherwise it clones it on the spot -- here before the queued product has written it)"""
    return None if t is None else t.view(t.shape)


def _wgrad_job_outputs(j):
    """fill in `out` / `db_out` of a job dict (fresh tensors where the caller named none); returns (dW, db or None)"""
    dc, a = j['dc'], j['a']
    Cin = a.shape[-1]
    KT = int(j.get('KT', 1))
    N = j.get('N') if j.get('N') is not None else dc.shape[-1]
    if j.get('out') is None:
        j['out'] = torch.empty((N, Cin, KT) if KT > 1 else (N, Cin), device=a.device, dtype=torch.float32)
    if j.get('with_db') and j.get('db_out') is None:
        j['db_out'] = torch.empty(N, device=a.device, dtype=torch.float32)
    return j['out'], (j['db_out'] if j.get('with_db') else None)


def gemm_wgrad_batch(jobs, later=False, params=()):
    """[gemm_wgrad(dc, a, KT, pad, Bn=, Tin=, Tout=, N=, out=, with_db=, db_out=) for each job dict] as ONE product launch and ONE slab-sum
    launch per 16 where the kernels allow (st_gemm_wgrad_batch: bit for bit the separate calls); returns [(dW, db or None)].
    later: inside a backward pass the jobs only join the queue (see above); pass `later=grad_first(*params)` and the same `params`."""
    from ._lib import StWgradJob
    if later:
        outs = [_wgrad_job_outputs(j) for j in jobs]
        if _queue_wgrads(jobs, params):
            return [(_alias(dw), _alias(db)) for dw, db in outs]
    lib = _lib.load()
    n = len(jobs)
    arr = (StWgradJob * n)()
    outs = []
    for i, j in enumerate(jobs):
        dc, a = j['dc'], j['a']
        if a.dim() == 3:
            Bn_, Tin_, Cin = a.shape
        else:
            Bn_, Tin_, Cin = 1, a.shape[0], a.shape[1]
        KT = int(j.get('KT', 1))
        Bn = j.get('Bn') or Bn_
        Tin = j.get('Tin') or Tin_
        Tout = j.get('Tout')
        if Tout is None:
            Tout = dc.shape[1] if dc.dim() == 3 else dc.shape[0] // Bn
        N = j.get('N') if j.get('N') is not None else dc.shape[-1]
        out = j.get('out')
        if out is None:
            out = torch.empty((N, Cin, KT) if KT > 1 else (N, Cin), device=a.device, dtype=torch.float32)
        db = None
        if j.get('with_db'):
            db = j.get('db_out')
            if db is None:
                db = torch.empty(N, device=a.device, dtype=torch.float32)
        q = arr[i]
        q.dC, q.lddc, q.dcoff, q.A, q.lda, q.dW, q.db = _p(dc), int(dc.stride(-2)), 0, _p(a), int(a.stride(-2)), _p(out), _p(db)
        q.Bn, q.Tin, q.Tout, q.Cin, q.N, q.KT, q.pad = int(Bn), int(Tin), int(Tout), int(Cin), int(N), KT, int(j.get('pad', 0))
        outs.append((out, db))
    ws = torch.empty(int(lib.st_gemm_wgrad_batch_workspace_floats(arr, n)), device=jobs[0]['a'].device, dtype=torch.float32)
    check(lib.st_gemm_wgrad_batch(arr, n, _p(ws), stream_handle()), 'st_gemm_wgrad_batch')
    return outs


def gemm_wgrad(dc, a, KT=1, pad=0, *, Bn=None, Tin=None, Tout=None, dcoff=0, N=None, pool_prev=False, out=None,
               accumulate=False, with_db=False, db_out=None, later=False, params=()):
    """dW (N, Cin[, KT]) of C = conv1d/linear(a, W): dc (Bn, Tout, >=dcoff+N) or (M, .), a (Bn, Tin, Cin) or (M, Cin).
    with_db: returns (dW, db) with db = the column sums of dc (the bias gradient), formed inside the same launches.
    out / db_out: where to write them (grad_slot of the parameters), else fresh tensors.
    later (= grad_first(*params), with the same `params`: the parameters whose gradients this writes): inside a backward pass the
    product may be queued and launched with others (flush_wgrads); the caller must not write to `dc` / `a` afterwards."""
    if later and WGRAD_DEFER and not pool_prev and not accumulate and dcoff == 0 and _graph_task() != -1:
        j = dict(dc=dc, a=a, KT=KT, pad=pad, Bn=Bn, Tin=Tin, Tout=Tout, N=N, out=out, with_db=with_db, db_out=db_out)
        dw, db = _wgrad_job_outputs(j)
        if _queue_wgrads([j], params):
            return (_alias(dw), _alias(db)) if with_db else _alias(dw)
        out, db_out = dw, db
    lib = _lib.load()
    if a.dim() == 3:
        Bn_, Tin_, Cin = a.shape
    else:
        Bn_, Tin_, Cin = 1, a.shape[0], a.shape[1]
    Bn = Bn or Bn_
    Tin = Tin or Tin_
    if Tout is None:
        Tout = dc.shape[1] if dc.dim() == 3 else dc.shape[0] // Bn
    N = N if N is not None else dc.shape[-1] - dcoff
    if out is None:
        out = torch.empty((N, Cin, KT) if KT > 1 else (N, Cin), device=a.device, dtype=torch.float32)
    nws = int(lib.st_gemm_wgrad_workspace_floats(int(Bn), int(Tout), int(Cin), int(N), int(KT)))
    ws = torch.empty(nws, device=a.device, dtype=torch.float32)
    if with_db:
        db = db_out if db_out is not None else torch.empty(N, device=a.device, dtype=torch.float32)
        check(lib.st_gemm_wgrad_db(_p(dc), int(dc.stride(-2)), int(dcoff), _p(a), int(a.stride(-2)), _p(out), _p(db), _p(ws), int(Bn),
                                   int(Tin), int(Tout), int(Cin), int(N), int(KT), int(pad), 1 if pool_prev else 0,
                                   1 if accumulate else 0, stream_handle()), 'st_gemm_wgrad_db')
        return out, db
    check(lib.st_gemm_wgrad(_p(dc), int(dc.stride(-2)), int(dcoff), _p(a), int(a.stride(-2)), _p(out), _p(ws), int(Bn),
                            int(Tin), int(Tout), int(Cin), int(N), int(KT), int(pad), 1 if pool_prev else 0,
                            1 if accumulate else 0, stream_handle()), 'st_gemm_wgrad')
    return out


def colsum(x2d, N=None, xoff=0, y2d=None, yoff=0, out=None, accumulate=False):
    M = x2d.shape[0]
    N = N if N is not None else x2d.shape[1] - xoff
    if out is None:
        out = torch.empty(N, device=x2d.device, dtype=torch.float32)
    check(_lib.load().st_colsum(_p(x2d), int(x2d.stride(0)), int(xoff), _p(y2d), int(y2d.stride(0)) if y2d is not None else 0,
                                int(yoff), M, N, _p(out), 1 if accumulate else 0, _p(_colreduce_ws(M, N, x2d.device)),
                                stream_handle()), 'st_colsum')
    return out


def act_bwd(dout2d, out2d, act, mask2d=None, dpre=None):
    M, N = dout2d.shape
    if dpre is None:
        dpre = torch.empty(M, N, device=dout2d.device, dtype=torch.float32)
    check(_lib.load().st_act_bwd(_p(dout2d), int(dout2d.stride(0)), _p(out2d), int(out2d.stride(0)) if out2d is not None else 0,
                                 ACT[act], _p(mask2d), int(mask2d.stride(0)) if mask2d is not None else 0, _p(dpre),
                                 int(dpre.stride(0)), M, N, stream_handle()), 'st_act_bwd')
    return dpre


def bn_bwd(dy2d, y2d, act, x2d, mean, var, w, eps, need_wb=True):
    """returns dx, dw, db for y = act(BN_train(x))"""
    M, N = x2d.shape
    dx = torch.empty(M, N, device=x2d.device, dtype=torch.float32)
    dw = torch.empty(N, device=x2d.device, dtype=torch.float32) if need_wb else None
    db = torch.empty(N, device=x2d.device, dtype=torch.float32) if need_wb else None
    ws = _colreduce_ws(M, N, x2d.device)
    check(_lib.load().st_bn_bwd(_p(dy2d), int(dy2d.stride(0)), 0, _p(y2d), int(y2d.stride(0)) if y2d is not None else 0, 0,
                                ACT[act], _p(x2d), int(x2d.stride(0)), 0, _p(mean), _p(var), _p(w), float(eps), M, N,
                                _p(dx), N, 0, _p(dw), _p(db), 0, _p(ws), stream_handle()), 'st_bn_bwd')
    return dx, dw, db


def bn_bwd_reduce(dy2d, y2d, act, x2d, mean, var, eps, mask2d=None):
    """-> s (2N): [sum dyb, sum dyb * xhat] over the local rows; mask2d: dy is multiplied by it on the way in (st_bn_norm_res_mask_fwd's backward)"""
    M, N = x2d.shape
    s = torch.empty(2 * N, device=x2d.device, dtype=torch.float32)
    ws = _colreduce_ws(M, N, x2d.device)
    if mask2d is not None:
        check(_lib.load().st_bn_bwd_reduce_masked(_p(dy2d), int(dy2d.stride(0)), _p(mask2d), int(mask2d.stride(0)), _p(y2d),
                                                  int(y2d.stride(0)) if y2d is not None else 0, ACT[act], _p(x2d), int(x2d.stride(0)),
                                                  _p(mean), _p(var), float(eps), M, N, _p(s), _p(ws), stream_handle()), 'st_bn_bwd_reduce_masked')
        return s
    check(_lib.load().st_bn_bwd_reduce(_p(dy2d), int(dy2d.stride(0)), 0, _p(y2d), int(y2d.stride(0)) if y2d is not None else 0, 0,
                                       ACT[act], _p(x2d), int(x2d.stride(0)), 0, _p(mean), _p(var), float(eps), M, N, _p(s), _p(ws),
                                       stream_handle()), 'st_bn_bwd_reduce')
    return s


def bn_bwd_apply(dy2d, y2d, act, x2d, mean, var, w, eps, s, Mstat, inv_total=None, mask2d=None, want_dres=None):
    """inv_total (device scalar, bn_sync_merge): s holds sums over all ranks' rows, divide them by the global count.
    want_dres is not None: the masked form -> (dx, dres or None) with dres = dy * mask (the gradient of a residual input)"""
    M, N = x2d.shape
    dx = torch.empty(M, N, device=x2d.device, dtype=torch.float32)
    if want_dres is not None:
        dres = torch.empty(M, N, device=x2d.device, dtype=torch.float32) if want_dres else None
        check(_lib.load().st_bn_bwd_apply_masked(_p(dy2d), int(dy2d.stride(0)), _p(mask2d), int(mask2d.stride(0)) if mask2d is not None else 0,
                                                 _p(y2d), int(y2d.stride(0)) if y2d is not None else 0, ACT[act], _p(x2d), int(x2d.stride(0)),
                                                 _p(mean), _p(var), _p(w), float(eps), M, N, _p(s), int(Mstat), _p(inv_total), _p(dx), N,
                                                 _p(dres), N, stream_handle()), 'st_bn_bwd_apply_masked')
        return dx, dres
    if inv_total is not None:
        check(_lib.load().st_bn_bwd_apply_sync(_p(dy2d), int(dy2d.stride(0)), 0, _p(y2d), int(y2d.stride(0)) if y2d is not None else 0, 0,
                                               ACT[act], _p(x2d), int(x2d.stride(0)), 0, _p(mean), _p(var), _p(w), float(eps), M, N,
                                               _p(s), _p(inv_total), _p(dx), N, 0, stream_handle()), 'st_bn_bwd_apply_sync')
        return dx
    check(_lib.load().st_bn_bwd_apply(_p(dy2d), int(dy2d.stride(0)), 0, _p(y2d), int(y2d.stride(0)) if y2d is not None else 0, 0,
                                      ACT[act], _p(x2d), int(x2d.stride(0)), 0, _p(mean), _p(var), _p(w), float(eps), M, N, _p(s),
                                      int(Mstat), _p(dx), N, 0, stream_handle()), 'st_bn_bwd_apply')
    return dx


def highway_fwd(H, Tg, x):
    y = torch.empty_like(x)
    check(_lib.load().st_highway_fwd(_p(H), _p(Tg), _p(x), _p(y), x.numel(), stream_handle()), 'st_highway_fwd')
    return y


def highway_bwd(dy, H, x, Tg):
    dH, dT, dx = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    check(_lib.load().st_highway_bwd(_p(dy), _p(H), _p(x), _p(Tg), _p(dH), _p(dT), _p(dx), x.numel(), stream_handle()),
          'st_highway_bwd')
    return dH, dT, dx


def _bn_bank_segs(xs, bns, means, vars_, dxs=None, sums=None, update_running=True):
    import ctypes as C
    n = len(xs)
    segs = (_lib.StBnBankSeg * n)()
    for k, (x, bn) in enumerate(zip(xs, bns)):
        sg = segs[k]
        sg.x, sg.ldx, sg.T = _p(x), int(x.stride(-2)), int(x.shape[1])
        sg.w, sg.b = _p(bn.weight), _p(bn.bias)
        if update_running and bn.running_mean is not None:
            sg.run_mean, sg.run_var = _p(bn.running_mean), _p(bn.running_var)
            sg.batches_tracked = _p(bn.num_batches_tracked, torch.int64)
        assert bn.momentum is not None, 'BatchNorm1d(momentum=None) (cumulative average) is not what the reference builds'
        sg.momentum, sg.eps = float(bn.momentum), float(bn.eps)
        sg.mean, sg.var = _p(means[k]), _p(vars_[k])
        if dxs is not None:
            sg.dx, sg.lddx, sg.sums = _p(dxs[k]), int(dxs[k].stride(-2)), _p(sums[k])
    return segs


def bn_bank_fwd(xs, bns, Tout):
    """K BatchNorm1d layers (batch statistics, running statistics updated) of K tensors (Bn, T_k, N) -> the bank (Bn, Tout, K N) and the
    (K, 2, N) statistics: 3 launches (st_bn_bank_fwd)"""
    lib = _lib.load()
    n, (Bn, _, N) = len(xs), xs[0].shape
    dev = xs[0].device
    stats = torch.empty(n, 2, N, device=dev, dtype=torch.float32)
    Y = torch.empty(Bn, Tout, n * N, device=dev, dtype=torch.float32)
    ws = torch.empty(int(lib.st_bn_bank_workspace_floats(n, int(Bn * max(x.shape[1] for x in xs)), int(N))), device=dev, dtype=torch.float32)
    segs = _bn_bank_segs(xs, bns, [stats[k, 0] for k in range(n)], [stats[k, 1] for k in range(n)])
    check(lib.st_bn_bank_fwd(segs, n, int(Bn), int(N), _p(Y), n * int(N), int(Tout), _p(ws), stream_handle()), 'st_bn_bank_fwd')
    return Y, stats


def bn_bank_fwd_sync(xs, bns, Tout):
    """bn_bank_fwd under SyncBN: this rank's records -> ONE all-gather -> merge -> normalise.  Returns the bank, the (K, 2, N) global
    statistics and inv_total (K) = 1 / global row count per segment."""
    from . import parallel
    lib = _lib.load()
    n, (Bn, _, N) = len(xs), xs[0].shape
    dev = xs[0].device
    stats = torch.empty(n, 2, N, device=dev, dtype=torch.float32)
    Y = torch.empty(Bn, Tout, n * N, device=dev, dtype=torch.float32)
    ws = torch.empty(int(lib.st_bn_bank_workspace_floats(n, int(Bn * max(x.shape[1] for x in xs)), int(N))), device=dev, dtype=torch.float32)
    segs = _bn_bank_segs(xs, bns, [stats[k, 0] for k in range(n)], [stats[k, 1] for k in range(n)])
    rec = torch.empty(n, 2 * N + 1, device=dev, dtype=torch.float32)
    check(lib.st_bn_bank_stats_record(segs, n, int(Bn), int(N), _p(rec), _p(ws), stream_handle()), 'st_bn_bank_stats_record')
    allrec = parallel.all_gather_(rec).contiguous()                       # (world, n, 2N + 1)
    inv_total = torch.empty(n, device=dev, dtype=torch.float32)
    check(lib.st_bn_bank_sync_merge(segs, n, int(Bn), int(N), _p(allrec), int(allrec.shape[0]), _p(inv_total), stream_handle()),
          'st_bn_bank_sync_merge')
    check(lib.st_bn_bank_norm(segs, n, int(Bn), int(N), _p(Y), n * int(N), int(Tout), stream_handle()), 'st_bn_bank_norm')
    return Y, stats, inv_total


def bn_bank_bwd_sync(dY, xs, bns, stats, inv_total, relu_in):
    """backward of bn_bank_fwd_sync: local sums -> ONE all-reduce -> apply.  Returns (dxs, LOCAL sums (K, 2, N)): the parameter gradients
    keep this rank's sums (the gradient reducer averages them over the ranks), dx uses the global ones."""
    from . import parallel
    lib = _lib.load()
    n, (Bn, _, N) = len(xs), xs[0].shape
    dev = xs[0].device
    dxs = [torch.empty_like(x) for x in xs]
    local = torch.empty(n, 2, N, device=dev, dtype=torch.float32)
    ws = torch.empty(int(lib.st_bn_bank_workspace_floats(n, int(Bn * max(x.shape[1] for x in xs)), int(N))), device=dev, dtype=torch.float32)
    means, vars_ = [stats[k, 0] for k in range(n)], [stats[k, 1] for k in range(n)]
    segs = _bn_bank_segs(xs, bns, means, vars_, dxs, [local[k] for k in range(n)], update_running=False)
    check(lib.st_bn_bank_bwd_reduce(segs, n, int(Bn), int(N), _p(dY), int(dY.stride(-2)), int(dY.shape[1]), _p(ws), stream_handle()),
          'st_bn_bank_bwd_reduce')
    glob = local.clone()
    parallel.all_reduce_sum_(glob)
    segs = _bn_bank_segs(xs, bns, means, vars_, dxs, [glob[k] for k in range(n)], update_running=False)
    check(lib.st_bn_bank_bwd_apply(segs, n, int(Bn), int(N), _p(dY), int(dY.stride(-2)), int(dY.shape[1]), 1 if relu_in else 0, _p(inv_total),
                                   stream_handle()), 'st_bn_bank_bwd_apply')
    return dxs, local


def bn_bank_bwd(dY, xs, bns, stats, relu_in):
    """backward of bn_bank_fwd: dx_k over every row of segment k (through the ReLU in front of the norm when relu_in) and the (K, 2, N)
    sums (d bias, d weight): 3 launches"""
    lib = _lib.load()
    n, (Bn, _, N) = len(xs), xs[0].shape
    dev = xs[0].device
    dxs = [torch.empty_like(x) for x in xs]
    sums = torch.empty(n, 2, N, device=dev, dtype=torch.float32)
    ws = torch.empty(int(lib.st_bn_bank_workspace_floats(n, int(Bn * max(x.shape[1] for x in xs)), int(N))), device=dev, dtype=torch.float32)
    segs = _bn_bank_segs(xs, bns, [stats[k, 0] for k in range(n)], [stats[k, 1] for k in range(n)], dxs, [sums[k] for k in range(n)],
                         update_running=False)
    check(lib.st_bn_bank_bwd(segs, n, int(Bn), int(N), _p(dY), int(dY.stride(-2)), int(dY.shape[1]), 1 if relu_in else 0, _p(ws),
                             stream_handle()), 'st_bn_bank_bwd')
    return dxs, sums


def cat_params(ws, transposed=False, pad_to=0):
    """parameters side by side as one GEMM operand (cached per weight version, refreshed with the other layouts in one launch)"""
    return _LAYOUTS.get_cat(list(ws), 1 if transposed else 0, pad_to)


def highway_ht_fwd(ht, x):
    M, Cn = x.shape
    y = torch.empty_like(x)
    check(_lib.load().st_highway_ht_fwd(_p(ht), _p(x), _p(y), int(M), int(Cn), stream_handle()), 'st_highway_ht_fwd')
    return y


def highway_ht_bwd(dy, ht, x):
    M, Cn = x.shape
    dht, dxd = torch.empty_like(ht), torch.empty_like(x)
    check(_lib.load().st_highway_ht_bwd(_p(dy), _p(ht), _p(x), _p(dht), _p(dxd), int(M), int(Cn), stream_handle()), 'st_highway_ht_bwd')
    return dht, dxd


def pool_prev_fwd(x):
    """MaxPool1d(2, stride 1, padding 1)(x)[:T] of a contiguous channels-last (B, T, C) tensor as a tensor of its own (C % 4 == 0)"""
    Bn, T, Cc = x.shape
    y = torch.empty_like(x)
    check(_lib.load().st_pool_prev_fwd(_p(x), _p(y), int(Bn), int(T), int(Cc), stream_handle()), 'st_pool_prev_fwd')
    return y


def pool_prev_bwd(dy_pooled, x):
    Bn, T, Cc = x.shape
    dx = torch.empty_like(x)
    check(_lib.load().st_pool_prev_bwd(_p(dy_pooled), _p(x), _p(dx), Bn, T, Cc, stream_handle()), 'st_pool_prev_bwd')
    return dx


def copy3d(dst, src, Bn, T, Cc, accumulate=False):
    """dst[b, t, :Cc] (+)= src[b, t, :Cc]; both (Bn, T, >=Cc) views with unit channel stride"""
    assert dst.stride(-1) == 1 and src.stride(-1) == 1
    check(_lib.load().st_copy3d(_p(dst), int(dst.stride(0)), int(dst.stride(1)), _p(src), int(src.stride(0)),
                                int(src.stride(1)), Bn, T, Cc, 1 if accumulate else 0, stream_handle()), 'st_copy3d')
    return dst


def scatter_add_rows(dout, idx, V):
    D = dout.shape[-1]
    idx = idx.contiguous()
    dtable = torch.empty(V, D, device=dout.device, dtype=torch.float32)
    fill_(dtable, 0.0)
    check(_lib.load().st_scatter_add_rows(_p(dout), _p(idx, torch.int64), _p(dtable), idx.numel(), D, V, stream_handle()),
          'st_scatter_add_rows')
    return dtable


# --------------------------------------------------------------------------------------------- packed operands
def t16_floats(B, K):
    return int(_lib.load().st_t16_floats(int(B), int(K)))


def pack_weight(ws, ks, N, lstm_H=0, ldws=None):
    """ws: list of (N, k_s) weight slices (views of torch weights); returns the P16 buffer"""
    lib = _lib.load()
    n = len(ws)
    karr = (C.c_int * n)(*[int(k) for k in ks])
    ldarr = (C.c_int * n)(*[int(ld) for ld in (ldws or [w.stride(0) for w in ws])])
    warr = (C.c_void_p * n)(*[_p(w) for w in ws])
    size = int(lib.st_packed_weight_floats(karr, n, int(N), int(lstm_H)))
    out = torch.empty(size, device=ws[0].device, dtype=torch.float32)
    check(lib.st_pack_weight(warr, ldarr, karr, n, int(N), int(lstm_H), _p(out), stream_handle()), 'st_pack_weight')
    return out


def pack_weight_t(ws):
    """P16 buffer of torch.cat(ws, 1).t() -- ws: list of (K, cols_s) matrices (row stride >= cols_s); no cat / transpose copy"""
    lib = _lib.load()
    n = len(ws)
    K = int(ws[0].shape[0])
    assert all(int(w.shape[0]) == K and w.stride(1) == 1 for w in ws)
    cols = [int(w.shape[1]) for w in ws]
    carr = (C.c_int * n)(*cols)
    ldarr = (C.c_int * n)(*[int(w.stride(0)) for w in ws])
    warr = (C.c_void_p * n)(*[_p(w) for w in ws])
    karr = (C.c_int * 1)(K)
    size = int(lib.st_packed_weight_floats(karr, 1, sum(cols), 0))
    out = torch.empty(size, device=ws[0].device, dtype=torch.float32)
    check(lib.st_pack_weight_t(warr, ldarr, carr, n, K, _p(out), stream_handle()), 'st_pack_weight_t')
    return out


def kb16(k):
    return (int(k) + 15) // 16


def t16_view(buf, kb_stride=None, kb0=0, K=None):
    """StT16View over a T16 buffer (kb_stride defaults to the k-blocks of a (B, K) buffer)"""
    v = _lib.StT16View()
    v.base = _p(buf)
    v.kb_stride = int(kb_stride if kb_stride is not None else kb16(K))
    v.kb0 = int(kb0)
    return v


def tile_rows(x, out=None, kb_stride=None, kb0=0):
    """natural (B, K) -> T16 (optionally into the k-block range kb0.. of a wider buffer)"""
    B, K = x.shape
    if out is None:
        out = torch.zeros(t16_floats(B, K), device=x.device, dtype=torch.float32)
    v = t16_view(out, kb_stride if kb_stride is not None else kb16(K), kb0)
    check(_lib.load().st_tile_rows(_p(x), int(x.stride(0)), C.byref(v), B, K, stream_handle()), 'st_tile_rows')
    return out


def untile_rows(x_t16, B, K, kb_stride=None, kb0=0):
    out = torch.empty(B, K, device=x_t16.device, dtype=torch.float32)
    v = t16_view(x_t16, kb_stride if kb_stride is not None else kb16(K), kb0)
    check(_lib.load().st_untile_rows(C.byref(v), _p(out), K, B, K, stream_handle()), 'st_untile_rows')
    return out


def untile_tape(flat, slots, Bp, kb_stride, segs):
    """T16 step tape (slots x [Bp/16][kb_stride][64][4]) -> natural (slots, Bp, sum(k)); segs = [(kb0, k), ...].
    Consecutive slots are consecutive batch tiles, so one launch per segment un-tiles every step."""
    width = sum(k for _, k in segs)
    out = torch.empty(slots, Bp, width, device=flat.device, dtype=torch.float32)
    merged = []          # segments that follow each other in whole k-blocks are one run in both layouts: one launch for the run
    for kb0, k in segs:
        if merged and merged[-1][1] % 16 == 0 and merged[-1][0] + merged[-1][1] // 16 == kb0:
            merged[-1] = (merged[-1][0], merged[-1][1] + k)
        else:
            merged.append((kb0, k))
    col = 0
    for kb0, k in merged:
        v = t16_view(flat, kb_stride, kb0)
        check(_lib.load().st_untile_rows(C.byref(v), _p(out) + 4 * col, width, slots * Bp, k, stream_handle()), 'st_untile_rows')
        col += k
    return out


def _vp(v):
    return C.byref(v) if v is not None else None


def lstm_cell_packed(packed_w, x_view, Kpad, b_ih, b_hh, c_prev, h_dst0, c_out, B, H, h_dst1=None, mask=None,
                     gates_out=None, ada_std=None, ada_mean=None, hadapt_dst=None):
    """x_view / *_dst: StT16View (see t16_view); Kpad = 16 * (k-blocks to reduce over)"""
    check(_lib.load().st_lstm_cell_packed_fwd(_p(packed_w), C.byref(x_view), int(Kpad), _p(b_ih), _p(b_hh),
                                              _p(c_prev), H, _p(mask), C.byref(h_dst0), _vp(h_dst1), _p(c_out), H,
                                              _p(gates_out), _p(ada_std), _p(ada_mean), _vp(hadapt_dst), int(B), int(H),
                                              stream_handle()), 'st_lstm_cell_packed_fwd')


def lstm_cell_packed_part(packed_w, w_kbs, x_view, Kpad, part, b_ih, b_hh, c_prev, h_dst0, c_out, B, H, h_dst1=None, mask=None, gates_out=None):
    """the cell over its LEADING Kpad columns + the slab `part` (B, 4H) of the products over the others (st_lstm_cell_packed_part_fwd);
    w_kbs = k-blocks per row tile of the whole packed matrix"""
    check(_lib.load().st_lstm_cell_packed_part_fwd(_p(packed_w), int(w_kbs), C.byref(x_view), int(Kpad), _p(part), _p(b_ih), _p(b_hh),
                                                   _p(c_prev), H, _p(mask), C.byref(h_dst0), _vp(h_dst1), _p(c_out), H, _p(gates_out),
                                                   int(B), int(H), stream_handle()), 'st_lstm_cell_packed_part_fwd')


def skinny_linear_packed(packed_w, x_view, Kpad, B, N, y=None, y_dst=None, bias=None, act=None, mask=None,
                         n_split=0, y2=None, rep=0, n_split2=0, act2=None, mask2=None, y3_dst=None):
    check(_lib.load().st_skinny_linear_packed_fwd(
        _p(packed_w), C.byref(x_view), int(Kpad), _p(bias), ACT[act], _p(mask),
        int(mask.stride(0)) if mask is not None else 0, _p(y), int(y.stride(0)) if y is not None else 0, _vp(y_dst),
        int(n_split), _p(y2), int(y2.stride(0)) if y2 is not None else 0, int(rep),
        int(n_split2), ACT[act2], _p(mask2), int(mask2.stride(0)) if mask2 is not None else 0, _vp(y3_dst),
        int(B), int(N), stream_handle()), 'st_skinny_linear_packed_fwd')


# --------------------------------------------------------------------------------------------- graphs
class Graph:
    """Records every st_* call issued inside `with g.capture():` on a private HIP stream into a
    hipGraph; `g.launch()` replays it on torch's current stream.  Every tensor touched inside the
    capture must stay alive as long as the graph is replayed: either allocated beforehand and kept
    by the caller, or allocated inside `with g.memory():` -- a private torch memory pool owned by
    this object, so that a buffer freed after the capture is never handed to anybody else (run the
    same code once inside `g.memory()` BEFORE the capture: the pool then serves the capture's
    allocations from its cache; a hipMalloc is not allowed while capturing)."""

    def __init__(self):
        self.lib = _lib.load()
        self.exec = None
        self._stream = C.c_void_p()
        check(self.lib.st_stream_create(C.byref(self._stream)), 'st_stream_create')
        self.pool = torch.cuda.MemPool()
        persist_status(torch.device('cuda', torch.cuda.current_device()))      # (allocated OUTSIDE the private pool: it outlives this graph)

    @contextmanager
    def memory(self):
        with torch.cuda.use_mem_pool(self.pool):
            yield

    @contextmanager
    def capture(self):
        torch.cuda.synchronize()
        check(self.lib.st_graph_begin(self._stream), 'st_graph_begin')
        ok = False
        try:
            with use_stream(self._stream.value):
                yield
            ok = True
        finally:
            ex = C.c_void_p()
            rc = self.lib.st_graph_end(self._stream, C.byref(ex))
            if ok:
                check(rc, 'st_graph_end')
                self.exec = ex

    def launch(self):
        check(self.lib.st_graph_launch(self.exec, stream_handle()), 'st_graph_launch')

    def __del__(self):
        try:
            if self.exec is not None:
                self.lib.st_graph_destroy(self.exec)
            self.lib.st_stream_destroy(self._stream)
        except Exception:
            pass


def device_info():
    lib = _lib.load()
    ncu, lds = C.c_int(), C.c_int()
    name = C.create_string_buffer(128)
    check(lib.st_device_info(C.byref(ncu), C.byref(lds), name, 128), 'st_device_info')
    return {'n_cu': ncu.value, 'lds_bytes': lds.value, 'name': name.value.decode()}
