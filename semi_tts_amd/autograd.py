"""Differentiable wrappers (torch.autograd.Function) around the HIP forward/backward kernels.

Training runs the same kernels as inference; torch.autograd only records
the graph and calls the `backward`s below, each of which is again HIP kernels of
libsemitts_hip.so (no torch arithmetic on activations; permute/flip of PARAMETERS is layout
plumbing).  What torch autograd derives for the reference's nn.Conv1d / nn.Linear /
nn.BatchNorm1d / Highway / MaxPool1d (src/module.py:421-431, :527-555, :597-611) is what these
functions compute; tests compare against CPU autograd through the oracle.
"""
import torch
from torch.autograd import Function

from . import ops


def _rows(t):
    return t.reshape(-1, t.shape[-1])


def _rows_strided(t):
    """(rows, N) view of a (..., N) tensor whose rows are evenly spaced in memory -- a column slice of a wider row-major buffer, e.g.
    the gradient of one block of a torch.cat over channels -- WITHOUT a copy (the kernels take a row stride); a copy otherwise"""
    if t.is_contiguous():
        return t.view(-1, t.shape[-1])
    if t.dim() >= 2 and t.stride(-1) == 1:
        ld = t.stride(-2)
        if ld >= t.shape[-1] and all(t.stride(i) == t.shape[i + 1] * t.stride(i + 1) for i in range(t.dim() - 2)):
            return t.as_strided((t.numel() // t.shape[-1], t.shape[-1]), (ld, 1), t.storage_offset())
    return t.contiguous().view(-1, t.shape[-1])


def _conv_grads(x, xw, w, b, dpre, pad, KT, N, Bn, Tin, To, pool_prev, pool_w, need_dx, need_dw, db_in_wgrad, dx_res=None):
    """input / weight (/ bias) gradients of a stride-1 channels-last conv from the gradient dpre (Bn, To, N) at its output; `dx_res`: a tensor
    added to dx in the product's epilogue (the gradient of a residual path around the layer).  The weight-gradient products are queued."""
    dx = dw = db = None
    if need_dx:
        wt, tap_major = ops.dx_weight(w.detach())        # (cached per weight version: all weights re-laid out by one launch per step)
        dx = ops.gemm(dpre, wt, Bn=Bn, Tin=To, Tout=Tin, pad=KT - 1 - pad, w_tap_major=tap_major, res=dx_res)
        if pool_prev:
            dx = ops.pool_prev_bwd(dx, x)
    # (the parameter gradients are written where ops.grad_slot says: under data parallelism, into their all-reduce bucket)
    if db_in_wgrad:
        dw, db = ops.gemm_wgrad(dpre, xw, KT, pad, Bn=Bn, Tin=Tin, Tout=To, N=N, pool_prev=pool_w, with_db=True,
                                out=ops.grad_slot(w), db_out=ops.grad_slot(b), later=ops.grad_first(w, b), params=(w, b))
        dw = dw.view(w.shape)
    elif need_dw:
        dw = ops.gemm_wgrad(dpre, xw, KT, pad, Bn=Bn, Tin=Tin, Tout=To, N=N, pool_prev=pool_w, out=ops.grad_slot(w), later=ops.grad_first(w),
                            params=(w,)).view(w.shape)
    return dx, dw, db


def _zero_stuff(dpre, Bn, To, N, Tin, pad, KT, stride):
    """a stride-s conv is the stride-1 conv sampled every s positions: its gradients are those of the stride-1 conv for an output gradient
    with zeros in between (only the speech encoder's second layer takes this path); returns (the stuffed gradient, its length)"""
    To1 = Tin + 2 * pad - KT + 1
    up = torch.empty(Bn, To1, N, device=dpre.device, dtype=torch.float32)
    ops.fill_(up, 0.0)
    ops.copy3d(up[:, 0:(To - 1) * stride + 1:stride], dpre.view(Bn, To, N), Bn, To, N)
    return up, To1


class _ConvFn(Function):
    """y = [act(conv1d/linear(pool?(x)) + b) (+ res)] (* mask), channels-last"""

    @staticmethod
    def forward(ctx, x, w, b, res, mask, pad, Tout, act, pool_prev, stride=1):
        assert not (res is not None and (act is not None or mask is not None)), 'residual only after a linear map'
        assert mask is None or act in (None, 'relu'), 'dropout mask is fused only after relu / identity'
        assert stride == 1 or (not pool_prev and Tout is None), 'strided conv: plain form only'
        x = x.contiguous()
        xp = None
        if pool_prev and x.dim() == 3 and x.shape[-1] % 4 == 0 and x.data_ptr() % 16 == 0:
            # the pooled input as a tensor of its own: forward AND weight-gradient products then run on the LDS-DMA kernels (which
            # cannot take a maximum on the way into LDS); x itself stays for the pool's backward
            xp = ops.pool_prev_fwd(x)
            y = ops.gemm(xp, w, pad=pad, stride=stride, Tout=Tout, bias=b, act_pre=act, res=res, mask=mask)
        else:
            y = ops.gemm(x, w, pad=pad, stride=stride, Tout=Tout, bias=b, act_pre=act, res=res, mask=mask, pool_prev=pool_prev)
        ctx.save_for_backward(x, w, y if act is not None else None, mask, xp, b)
        ctx.cfg = (pad, act, pool_prev, b is not None, res is not None, stride)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, mask, xp, b = ctx.saved_tensors
        pad, act, pool_prev, has_b, has_res, stride = ctx.cfg
        xw, pool_w = (xp, False) if xp is not None else (x, pool_prev)       # the weight gradient's activation operand
        dy = dy.contiguous()
        N = w.shape[0]
        KT = w.shape[2] if w.dim() == 3 else 1
        if act is not None or mask is not None:
            dpre = ops.act_bwd(_rows(dy), _rows(y) if y is not None else None, act,
                               _rows(mask) if mask is not None else None).view(dy.shape)
        else:
            dpre = dy
        dx = dw = db = None
        if x.dim() == 3:
            Bn, Tin, Cin = x.shape
            To = dy.shape[1]
        else:
            Bn, Tin, Cin = 1, x.shape[0], x.shape[1]
            To = dy.shape[0]
        # (the bias gradient rides in the weight-gradient launches when both are wanted and the conv has stride 1)
        db_in_wgrad = has_b and ctx.needs_input_grad[2] and ctx.needs_input_grad[1] and stride == 1
        if has_b and ctx.needs_input_grad[2] and not db_in_wgrad:
            db = ops.colsum(_rows(dpre), out=ops.grad_slot(b))
        if stride > 1:
            dpre, To = _zero_stuff(dpre, Bn, To, N, Tin, pad, KT, stride)
        if (KT == 1 and w.dim() == 2 and pad == 0 and N % 4 != 0 and N >= 64 and Cin % 4 == 0 and not pool_prev and stride == 1 and w.is_contiguous()
                and _rows(dpre).shape[0] >= 1024 and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]):
            # a Linear whose output width is not a multiple of 4 (Linear(160, 1025) of the postnet): rows of dy are not 16-byte addressable
            # and both backward products fall to the element-wise kernels (68 + 74 us at C2).  One copy into rows padded to Np floats
            # (pad columns zero) and the padded transposed weight put them on the LDS-DMA kernels (~20 + 25 us).
            Np = (N + 3) // 4 * 4
            d2 = _rows(dpre)
            M_ = d2.shape[0]
            dyp = torch.empty(M_, Np, device=d2.device, dtype=torch.float32)
            ops.copy2d(dyp, d2, M_, N)
            dyp[:, N:].zero_()
            dx = ops.gemm(dyp, ops.cat_params([w.detach()], transposed=True, pad_to=Np)).view(x.shape)
            x2 = _rows(xw)
            if has_b and ctx.needs_input_grad[2]:
                dwp, dbp = ops.gemm_wgrad(dyp, x2, with_db=True, later=ops.grad_first(w, b), params=(w, b))
                db = dbp[:N]
            else:
                dwp = ops.gemm_wgrad(dyp, x2, later=ops.grad_first(w), params=(w,))
            dw = dwp[:N]
            dres = dy if has_res and ctx.needs_input_grad[3] else None
            return dx, dw, db, dres, None, None, None, None, None, None
        dx, dw, db2 = _conv_grads(x, xw, w, b, dpre, pad, KT, N, Bn, Tin, To, pool_prev, pool_w, ctx.needs_input_grad[0],
                                  ctx.needs_input_grad[1], db_in_wgrad)
        if db_in_wgrad:
            db = db2
        dres = dy if has_res and ctx.needs_input_grad[3] else None
        return dx, dw, db, dres, None, None, None, None, None, None


class _ConvLayerFn(Function):
    """ConvLayer.forward in training (src/module.py:638-648) as ONE autograd node: Conv1d (+ bias) -> BatchNorm1d over the batch ->
    activation -> (+ x) -> dropout mask.  Forward: the conv product, the statistics (two launches), ONE launch for normalisation +
    activation + residual + mask (the reference: BatchNorm apply, tanh, add, bernoulli_, div_, mul).  Backward: the two halves of the
    BatchNorm backward take the mask on the way in and hand out the residual's gradient, which the input-gradient product adds in its
    epilogue (no mask / add / accumulate launches).  Under parallel.sync_batchnorm the statistics are the global batch's (_BnTrainFn)."""

    @staticmethod
    def forward(ctx, x, w, b, bn_w, bn_b, run_mean, run_var, tracked, mask, pad, stride, act, residual, eps, momentum):
        from . import parallel
        x = x.contiguous()
        y = ops.gemm(x, w, pad=pad, stride=stride, bias=b)
        N = y.shape[-1]
        y2 = _rows(y)
        M = y2.shape[0]
        sync = parallel.sync_bn_active()
        inv_total = None
        if sync:
            parallel._COUNTS['syncbn_fwd'] += 1
            rec = parallel.all_gather_(ops.bn_stats_record(y2, 0, N, tracked))
            mean, var, inv_total = ops.bn_sync_merge(rec, N, run_mean, run_var, momentum)
        else:
            mean, var = ops.bn_stats(y2, 0, N, run_mean, run_var, momentum, tracked)
        t, out = ops.bn_norm_res_mask(y2, mean, var, bn_w, bn_b, eps, act, _rows(x) if residual else None,
                                      _rows(mask) if mask is not None else None, want_t=act is not None)
        ctx.save_for_backward(x, w, b, y, t, mean, var, bn_w, mask, inv_total)
        ctx.cfg = (pad, stride, act, residual, eps, M, sync)
        return out.view(y.shape)

    @staticmethod
    def backward(ctx, dout):
        from . import parallel
        x, w, b, y, t, mean, var, bn_w, mask, inv_total = ctx.saved_tensors
        pad, stride, act, residual, eps, M, sync = ctx.cfg
        d2, y2 = _rows_strided(dout), _rows(y)
        N = y2.shape[1]
        m2 = _rows(mask) if mask is not None else None
        s = ops.bn_bwd_reduce(d2, t, act, y2, mean, var, eps, mask2d=m2)
        dbn_b, dbn_w = s[:N], s[N:]
        if sync:
            dbn_b, dbn_w = dbn_b.clone(), dbn_w.clone()
            parallel._COUNTS['syncbn_bwd'] += 1
            parallel.all_reduce_sum_(s)
        dpre, dres = ops.bn_bwd_apply(d2, t, act, y2, mean, var, bn_w, eps, s, M, inv_total, mask2d=m2, want_dres=residual)
        Bn, Tin, Cin = x.shape
        To = y.shape[1]
        KT = w.shape[2]
        dpre = dpre.view(Bn, To, N)
        db = None
        has_b = b is not None
        db_in_wgrad = has_b and ctx.needs_input_grad[2] and ctx.needs_input_grad[1] and stride == 1
        if has_b and ctx.needs_input_grad[2] and not db_in_wgrad:
            db = ops.colsum(_rows(dpre), out=ops.grad_slot(b))
        if stride > 1:
            dpre, To = _zero_stuff(dpre, Bn, To, N, Tin, pad, KT, stride)
        dx, dw, db2 = _conv_grads(x, x, w, b, dpre, pad, KT, N, Bn, Tin, To, False, False, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                  db_in_wgrad, dx_res=dres.view(x.shape) if dres is not None else None)
        if db_in_wgrad:
            db = db2
        if dx is None and dres is not None:
            dx = dres.view(x.shape)
        return dx, dw, db, dbn_w, dbn_b, None, None, None, None, None, None, None, None, None, None


def conv_layer(x, conv, bn, mask, pad, stride, act, residual):
    """differentiable ConvLayer (speech encoder): see _ConvLayerFn"""
    return _ConvLayerFn.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, mask,
                              pad, stride, act, residual, bn.eps, bn.momentum)


class _MaskMulFn(Function):
    """y = x * mask (a scaled dropout mask drawn elsewhere); backward dy * mask -- the library's strided multiply, no ATen launch"""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        x = x.contiguous()
        return ops.act_bwd(_rows(x), None, None, _rows(mask)).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        mask, = ctx.saved_tensors
        return ops.act_bwd(_rows_strided(dy), None, None, _rows(mask)).view(dy.shape), None


def mask_mul(x, mask):
    return _MaskMulFn.apply(x, mask)


def _contig3(x):
    """x (B, T, D) as a contiguous tensor: itself, or one strided copy by the library (no ATen launch)"""
    if x.is_contiguous():
        return x
    if x.dim() == 3 and x.stride(-1) == 1:
        out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
        return ops.copy3d(out, x, x.shape[0], x.shape[1], x.shape[2])
    return x.contiguous()


class _PaddedConcatFn(Function):
    """VQVAE.padded_concat (src/vqvae.py:259-271): two (B, T, D) batches stacked on the batch axis, the shorter one zero-padded in time.
    One fill (only when something is padded) + two strided copies of the library; backward hands out the two blocks of the gradient."""

    @staticmethod
    def forward(ctx, a, b):
        Ba, Ta, D = a.shape
        Bb, Tb, _ = b.shape
        T = max(Ta, Tb)
        out = torch.empty(Ba + Bb, T, D, device=a.device, dtype=torch.float32)
        if Ta != Tb:
            ops.fill_(out, 0.0)
        ops.copy3d(out[:Ba], a if a.stride(-1) == 1 else a.contiguous(), Ba, Ta, D)
        ops.copy3d(out[Ba:], b if b.stride(-1) == 1 else b.contiguous(), Bb, Tb, D)
        ctx.dims = (Ba, Ta, Bb, Tb)
        return out

    @staticmethod
    def backward(ctx, dout):
        Ba, Ta, Bb, Tb = ctx.dims
        da = _contig3(dout[:Ba, :Ta]) if ctx.needs_input_grad[0] else None
        db = _contig3(dout[Ba:, :Tb]) if ctx.needs_input_grad[1] else None
        return da, db


def padded_concat(a, b):
    return _PaddedConcatFn.apply(a, b)


class _SplitPairFn(Function):
    """(x[:Bp, :Tp], x[Bp:, :Tu]) of a (B, T, D) tensor -- the unpacking of VQVAE.text_to_speech (src/vqvae.py:187-195) -- with a backward
    that writes the two incoming gradients into ONE tensor (two strided copies of the library; a fill only when a block is shorter than T).
    Plain slicing costs autograd two full-size zero fills, two copies and an add (68 MB each for the linear spectrogram at B = 64)."""

    @staticmethod
    def forward(ctx, x, Bp, Tp, Tu):
        ctx.dims = (tuple(x.shape), Bp, Tp, Tu)
        ctx.set_materialize_grads(False)
        return x[:Bp, :Tp], x[Bp:, :Tu]

    @staticmethod
    def backward(ctx, da, db):
        shape, Bp, Tp, Tu = ctx.dims
        B, T, D = shape
        if da is None and db is None:
            return None, None, None, None
        dev = (da if da is not None else db).device
        dx = torch.empty(shape, device=dev, dtype=torch.float32)
        if Tp != T or Tu != T or da is None or db is None:
            ops.fill_(dx, 0.0)
        if da is not None:
            ops.copy3d(dx[:Bp], da if da.stride(-1) == 1 else da.contiguous(), Bp, Tp, D)
        if db is not None:
            ops.copy3d(dx[Bp:], db if db.stride(-1) == 1 else db.contiguous(), B - Bp, Tu, D)
        return dx, None, None, None


def split_pair(x, Bp, Tp, Tu):
    T = x.shape[1]
    return _SplitPairFn.apply(x, int(Bp), min(int(Tp), T), min(int(Tu), T))


class _ScalarCombineFn(Function):
    """total = sum_i w_i x_i over one-element tensors (+ up to three more weighted sums for the log) in ONE launch; backward: one launch
    for all the w_i * dtotal.  The reference's `total_loss = total_loss + w * loss` chain (bin/train_vqvae.py:208-233) is a one-element torch
    kernel per operator, forward and backward."""

    @staticmethod
    def forward(ctx, W, *xs):
        from . import _lib
        import ctypes as C
        n, m = len(xs), len(W)
        dev = xs[0].device
        outs = [torch.empty((), device=dev, dtype=torch.float32) for _ in range(m)]
        xa = (C.c_void_p * n)(*[ops._p(x) for x in xs])
        oa = (C.c_void_p * m)(*[ops._p(o) for o in outs])
        wa = (C.c_float * (n * m))(*[float(v) for row in W for v in row])
        _lib.check(_lib.load().st_scalar_combine(xa, n, wa, m, oa, ops.stream_handle()), 'st_scalar_combine')
        ctx.w = [float(v) for v in W[0]]
        ctx.mark_non_differentiable(*outs[1:])
        return tuple(outs)

    @staticmethod
    def backward(ctx, dtotal, *_):
        from . import _lib
        import ctypes as C
        n = len(ctx.w)
        out = torch.empty(n, device=dtotal.device, dtype=torch.float32)
        wa = (C.c_float * n)(*ctx.w)
        _lib.check(_lib.load().st_scalar_fanout(ops._p(dtotal.contiguous()), wa, n, ops._p(out), ops.stream_handle()), 'st_scalar_fanout')
        return (None,) + tuple(out[i] if ctx.needs_input_grad[i + 1] else None for i in range(n))


def scalar_combine(W, xs):
    """W: rows of weights (row 0 = the differentiable total); xs: one-element tensors.  -> tuple of len(W) 0-d tensors"""
    return _ScalarCombineFn.apply([list(r) for r in W], *xs)


class _ConvGroupFn(Function):
    """n convolutions / linear maps of the SAME input (the K convs of the CBHG bank, src/module.py:590-598; the two directions' input
    projections of a bidirectional GRU / LSTM): y_k = act(conv1d_k(x) + b_k).  Forward: the n products go out as ONE launch where the
    kernels allow (ops.gemm_flush, as in inference).  Backward: the input gradient is accumulated by the products themselves (product
    k adds product k - 1's result in its epilogue) instead of n tensors that autograd then adds up with n - 1 elementwise launches.
    Arguments: n, act, x, then n weights, n biases (None = no bias), n (pad, Tout) pairs."""

    @staticmethod
    def forward(ctx, n, act, x, *args):
        ws, bs, cfg = args[:n], args[n:2 * n], args[2 * n:2 * n + 2 * n]
        grad_done = bool(args[4 * n]) if len(args) > 4 * n else False     # the incoming gradients are already at the pre-activations
        x = x.contiguous()
        jobs, ys = [], []
        for k in range(n):
            ys.append(ops.gemm(x, ws[k], pad=int(cfg[2 * k]), Tout=cfg[2 * k + 1], bias=bs[k], act_pre=act, collect=jobs))
        ops.gemm_flush(jobs)
        if grad_done:
            act = None
        ctx.save_for_backward(x, *ws, *bs, *(ys if act is not None else []))
        ctx.cfg = (n, act, [int(cfg[2 * k]) for k in range(n)], [b is not None for b in bs])
        ctx.extra = len(args) - 4 * n
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        n, act, pads, has_b = ctx.cfg
        sv = ctx.saved_tensors
        x, ws, bs, ys = sv[0], sv[1:1 + n], sv[1 + n:1 + 2 * n], sv[1 + 2 * n:]
        if x.dim() == 3:
            Bn, Tin, Cin = x.shape
        else:
            Bn, Tin, Cin = 1, x.shape[0], x.shape[1]
        dx, dws, dbs = None, [None] * n, [None] * n
        wjobs = []                                       # the n weight-gradient products go out together: one launch + one slab sum
        for k in range(n):
            w = ws[k]
            N, KT = w.shape[0], (w.shape[2] if w.dim() == 3 else 1)
            dy = dys[k].contiguous()
            dpre = ops.act_bwd(_rows(dy), _rows(ys[k]), act, None).view(dy.shape) if act is not None else dy
            To = dy.shape[1] if x.dim() == 3 else dy.shape[0]
            if ctx.needs_input_grad[2]:
                wt, tap_major = ops.dx_weight(w.detach())
                if dx is None:
                    dx = ops.gemm(dpre, wt, Bn=Bn, Tin=To, Tout=Tin, pad=KT - 1 - pads[k], w_tap_major=tap_major)
                else:                                    # (in place: every element is read and written by the same thread)
                    ops.gemm(dpre, wt, dx, Bn=Bn, Tin=To, Tout=Tin, pad=KT - 1 - pads[k], w_tap_major=tap_major, res=dx)
            want_w, want_b = ctx.needs_input_grad[3 + k], has_b[k] and ctx.needs_input_grad[3 + n + k]
            if want_w:       # (the parameter gradients are written where ops.grad_slot says)
                wjobs.append((k, dict(dc=dpre, a=x, KT=KT, pad=pads[k], Bn=Bn, Tin=Tin, Tout=To, N=N, out=ops.grad_slot(w),
                                      with_db=bool(want_b), db_out=ops.grad_slot(bs[k]) if want_b else None)))
            elif want_b:
                dbs[k] = ops.colsum(_rows(dpre), out=ops.grad_slot(bs[k]))
        if wjobs:
            wpar = tuple(ws[k] for k, _ in wjobs) + tuple(bs[k] for k, j in wjobs if j['with_db'])
            for (k, _), (dw, db) in zip(wjobs, ops.gemm_wgrad_batch([j for _, j in wjobs], later=ops.grad_first(*wpar), params=wpar)):
                dws[k], dbs[k] = dw.view(ws[k].shape), db
        if dx is not None:
            dx = dx.view(x.shape)
        return (None, None, dx) + tuple(dws) + tuple(dbs) + (None,) * (2 * n + ctx.extra)


def conv_group(x, weights, pads, Touts, act=None, biases=None, act_grad_done=False):
    """[act(conv1d_k(x) + b_k)] for n convolutions / linear maps of one input (see _ConvGroupFn).  act_grad_done: whoever consumes the
    outputs returns gradients that already went through act's derivative (batch_norm_bank with relu_in)."""
    n = len(weights)
    biases = list(biases) if biases is not None else [None] * n
    cfg = []
    for p_, t_ in zip(pads, Touts):
        cfg += [int(p_), t_]
    extra = (True,) if act_grad_done else ()
    return list(_ConvGroupFn.apply(n, act, x, *weights, *biases, *cfg, *extra))


def conv(x, w, b=None, *, pad=0, Tout=None, act=None, res=None, mask=None, pool_prev=False, stride=1):
    """differentiable ops.gemm (conv1d over channels-last rows, or linear when w is 2-D)"""
    return _ConvFn.apply(x, w, b, res, mask, pad, Tout, act, pool_prev, stride)


def linear(x, w, b=None, act=None, mask=None):
    lead = x.shape[:-1]
    y = _ConvFn.apply(x.reshape(-1, x.shape[-1]), w, b, None,
                      mask.reshape(-1, mask.shape[-1]) if mask is not None else None, 0, None, act, False)
    return y.view(*lead, -1)


class _BnTrainFn(Function):
    """y = act(BatchNorm1d(x)) with batch statistics over all leading dims (training mode);
    updates running_mean / running_var in place like nn.BatchNorm1d (unbiased running var).
    With parallel.sync_batchnorm() under torch.distributed the statistics are those of the global batch."""

    @staticmethod
    def forward(ctx, x, weight, bias, run_mean, run_var, eps, momentum, act, batches_tracked=None):
        from . import parallel
        x = x.contiguous()
        N = x.shape[-1]
        x2 = _rows(x)
        M = x2.shape[0]
        sync = parallel.sync_bn_active()
        inv_total = None
        if sync:
            # ONE collective: every rank's (mean, M2, count) record, written by the statistics launch itself and merged exactly and
            # identically on every rank in rank order (Chan et al.) by one more launch.  The counts stay on the device -- no host
            # round trip, and shards of different sizes (a ragged last batch, B % world != 0, different T per rank in the speech
            # encoder) are weighted by their true row counts.
            parallel._COUNTS['syncbn_fwd'] += 1
            rec = parallel.all_gather_(ops.bn_stats_record(x2, 0, N, batches_tracked))      # (world, 2N + 1)
            mean, var, inv_total = ops.bn_sync_merge(rec, N, run_mean, run_var, momentum)   # inv_total: 1 / the GLOBAL row count
        else:
            mean, var = ops.bn_stats(x2, 0, N, run_mean, run_var, momentum, batches_tracked)
        y = ops.bn_norm(x2, 0, N, mean, var, weight, bias, eps, act).view(x.shape)
        ctx.save_for_backward(x, y if act is not None else None, mean, var, weight, inv_total)
        ctx.cfg = (eps, act, M, sync)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import parallel
        x, y, mean, var, weight, inv_total = ctx.saved_tensors
        eps, act, M, sync = ctx.cfg
        dy2, y2, x2 = _rows_strided(dy), _rows(y) if y is not None else None, _rows(x)      # (dy: often a column slice of a cat's gradient)
        N = x2.shape[1]
        s = ops.bn_bwd_reduce(dy2, y2, act, x2, mean, var, eps)
        db, dw = s[:N], s[N:]                           # parameter gradients: local sums (averaged over ranks later)
        if sync:
            db, dw = db.clone(), dw.clone()             # (s is all-reduced in place below; otherwise the views are handed out as they are)
            parallel._COUNTS['syncbn_bwd'] += 1
            parallel.all_reduce_sum_(s)
        dx = ops.bn_bwd_apply(dy2, y2, act, x2, mean, var, weight, eps, s, M, inv_total)     # sync: the sums / the GLOBAL row count
        return dx.view(x.shape), dw, db, None, None, None, None, None, None


def batch_norm_train(x, bn, act=None):
    """(num_batches_tracked is incremented by the statistics launch itself)"""
    return _BnTrainFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, act, bn.num_batches_tracked)


class _BnTrainGroupFn(Function):
    """_BnTrainFn for n INDEPENDENT layers of equal width at once (the K BatchNorms of the CBHG conv bank, src/module.py:590-598: they
    read one input and none depends on another).  Without torch.distributed it is n times the single-layer function; under SyncBN the
    n layers share their collectives -- ONE all-gather of the n (mean, M2, count) records in forward, ONE all-reduce of the n pairs of
    sums in backward (8 + 8 collectives of ~43 us each -> 1 + 1 per step).  Arguments: n, act, then n x (x, weight, bias, run_mean,
    run_var, batches_tracked), then n x (eps, momentum) as floats."""

    @staticmethod
    def forward(ctx, n, act, *args):
        from . import parallel
        tens, scal = args[:6 * n], args[6 * n:]
        xs = [tens[6 * i].contiguous() for i in range(n)]
        N = xs[0].shape[-1]
        assert all(x.shape[-1] == N for x in xs), 'group BatchNorm: equal channel counts'
        sync = parallel.sync_bn_active()
        stats = []
        if sync:
            parallel._COUNTS['syncbn_fwd'] += 1
            recs = torch.cat([ops.bn_stats_record(_rows(xs[i]), 0, N, tens[6 * i + 5]) for i in range(n)])       # (n (2N + 1),)
            allr = parallel.all_gather_(recs)                                                                     # (world, n (2N + 1))
            per = allr.view(allr.shape[0], n, 2 * N + 1).transpose(0, 1).contiguous()                             # (n, world, 2N + 1)
            for i in range(n):
                stats.append(ops.bn_sync_merge(per[i], N, tens[6 * i + 3], tens[6 * i + 4], scal[2 * i + 1]))
        else:
            for i in range(n):
                stats.append(ops.bn_stats(_rows(xs[i]), 0, N, tens[6 * i + 3], tens[6 * i + 4], scal[2 * i + 1], tens[6 * i + 5]) + (None,))
        ys, saved = [], []
        for i in range(n):
            mean, var, inv_total = stats[i]
            y = ops.bn_norm(_rows(xs[i]), 0, N, mean, var, tens[6 * i + 1], tens[6 * i + 2], scal[2 * i], act).view(xs[i].shape)
            ys.append(y)
            saved += [xs[i], y if act is not None else None, mean, var, tens[6 * i + 1], inv_total]
        ctx.save_for_backward(*saved)
        ctx.cfg = (n, act, sync, [float(scal[2 * i]) for i in range(n)], [_rows(x).shape[0] for x in xs])
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        from . import parallel
        n, act, sync, eps, Ms = ctx.cfg
        sv = ctx.saved_tensors
        N = sv[0].shape[-1]
        sums = []
        for i in range(n):
            x, y, mean, var = sv[6 * i], sv[6 * i + 1], sv[6 * i + 2], sv[6 * i + 3]
            sums.append(ops.bn_bwd_reduce(_rows_strided(dys[i]), _rows(y) if y is not None else None, act, _rows(x), mean, var, eps[i]))
        if sync:
            local = torch.cat(sums)                      # the parameter gradients keep the LOCAL sums (averaged over ranks later)
            flat = local.clone()
            parallel._COUNTS['syncbn_bwd'] += 1
            parallel.all_reduce_sum_(flat)
            glob = [flat[2 * N * i:2 * N * (i + 1)] for i in range(n)]
            sums = [local[2 * N * i:2 * N * (i + 1)] for i in range(n)]
        else:
            glob = sums
        grads = []
        for i in range(n):
            x, y, mean, var, weight, inv_total = sv[6 * i:6 * i + 6]
            dx = ops.bn_bwd_apply(_rows_strided(dys[i]), _rows(y) if y is not None else None, act, _rows(x), mean, var, weight, eps[i], glob[i], Ms[i], inv_total)
            grads += [dx.view(x.shape), sums[i][N:], sums[i][:N], None, None, None]
        return (None, None) + tuple(grads) + (None,) * (2 * n)


class _BnBankFn(Function):
    """The K BatchNorm1d layers of the CBHG conv bank (src/module.py:590-598), straight into the concatenated bank: 3
    launches forward (K x 3 + a torch.cat before), 3 backward (K x (reduce, merge, apply, ReLU backward) + the zero-padding of the four
    trimmed positions before).  relu_in: the inputs are relu(conv) and the ReLU's backward rides in the apply launch -- the gradients
    returned for xs are then the gradients at the convs' PRE-activations (conv_group(..., act_grad_done=True) takes them as such)."""

    @staticmethod
    def forward(ctx, holder, Tout, relu_in, *tens):
        from . import parallel
        n = len(holder)
        xs = [t.contiguous() for t in tens[:n]]
        ctx.sync = parallel.sync_bn_active()
        if ctx.sync:       # SyncBN: the same launches in stages, ONE all-gather of the K records between them
            parallel._COUNTS['syncbn_fwd'] += 1
            Y, stats, inv_total = ops.bn_bank_fwd_sync(xs, holder, Tout)
        else:
            Y, stats = ops.bn_bank_fwd(xs, holder, Tout)
            inv_total = stats.new_empty(0)
        ctx.holder, ctx.relu_in = holder, relu_in
        ctx.save_for_backward(stats, inv_total, *xs)
        return Y

    @staticmethod
    def backward(ctx, dY):
        from . import parallel
        sv = ctx.saved_tensors
        stats, inv_total, xs = sv[0], sv[1], list(sv[2:])
        n = len(xs)
        if ctx.sync:
            parallel._COUNTS['syncbn_bwd'] += 1
            dxs, sums = ops.bn_bank_bwd_sync(dY.contiguous(), xs, ctx.holder, stats, inv_total, ctx.relu_in)
        else:
            dxs, sums = ops.bn_bank_bwd(dY.contiguous(), xs, ctx.holder, stats, ctx.relu_in)
        gw = [sums[k, 1] for k in range(n)]
        gb = [sums[k, 0] for k in range(n)]
        return (None, None, None) + tuple(dxs) + tuple(gw) + tuple(gb)


def batch_norm_bank(xs, bns, Tout, relu_in):
    """cat_k BatchNorm1d_k(x_k)[:, :Tout] (batch statistics; no SyncBN)"""
    bns = list(bns)
    return _BnBankFn.apply(bns, Tout, relu_in, *xs, *[bn.weight for bn in bns], *[bn.bias for bn in bns])


def batch_norm_train_group(xs, bns, act=None):
    """act(BatchNorm1d_i(x_i)) of n independent layers of equal width with shared SyncBN collectives (_BnTrainGroupFn)"""
    n = len(xs)
    args = []
    for x, bn in zip(xs, bns):
        args += [x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked]
    for bn in bns:
        args += [float(bn.eps), float(bn.momentum)]
    return list(_BnTrainGroupFn.apply(n, act, *args))


class _LayerNormFn(Function):
    """nn.LayerNorm over the last dimension (the speech encoder's layer_norm, src/asr.py:38-39,58; the normalised prenet Linear,
    src/module.py:508-521)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        x2 = _rows(x)
        y, mean, rstd = ops.layer_norm(x2, gamma, beta, eps, want_stats=True)
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dy2 = _rows(dy.contiguous())
        dx, dyxhat = ops.layer_norm_bwd(dy2, _rows(x), gamma, mean, rstd)
        return dx.view(x.shape), ops.colsum(dyxhat), ops.colsum(dy2), None


def layer_norm(x, ln):
    """ln: nn.LayerNorm over the last dimension"""
    return _LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps)


class _LogSoftmaxFn(Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.log_softmax(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        return ops.log_softmax_bwd(dy, y)


def log_softmax(x):
    return _LogSoftmaxFn.apply(x)


class _HighwayFn(Function):
    """y = H*T + x*(1-T)      ref: src/module.py:551-554"""

    @staticmethod
    def forward(ctx, H, Tg, x):
        H, Tg, x = H.contiguous(), Tg.contiguous(), x.contiguous()
        ctx.save_for_backward(H, Tg, x)
        return ops.highway_fwd(H, Tg, x)

    @staticmethod
    def backward(ctx, dy):
        H, Tg, x = ctx.saved_tensors
        return ops.highway_bwd(dy.contiguous(), H, x, Tg)


def highway_combine(H, Tg, x):
    return _HighwayFn.apply(H, Tg, x)


class _HighwayLayerFn(Function):
    """One Highway layer (src/module.py:541-555) with its two Linear layers as ONE product: [h | t] = x [W_H ; W_T]^T + [b_H | b_T]
    (N = 2C), y = relu(h) sigmoid(t) + x (1 - sigmoid(t)).  Forward 2 launches (3 as two convs + combine), backward 4 (11): the
    gradients at the two pre-activations leave one kernel side by side and feed one input-gradient product (K = 2C, the direct
    path dy (1 - T) added in its epilogue) and one weight-gradient product (the two bias gradients in its launches)."""

    @staticmethod
    def forward(ctx, x, Hw, Hb, Tw, Tb):
        lead = x.shape[:-1]
        x2 = x.contiguous().view(-1, x.shape[-1])
        ht = ops.gemm(x2, ops.cat_params([Hw, Tw]), bias=ops.cat_params([Hb, Tb]))
        ctx.save_for_backward(x2, ht, Hw, Tw)
        ctx.lead = lead
        ctx.hb = (Hb, Tb)                # (identity only: which parameters the bias gradients belong to)
        return ops.highway_ht_fwd(ht, x2).view(*lead, -1)

    @staticmethod
    def backward(ctx, dy):
        x2, ht, Hw, Tw = ctx.saved_tensors
        Cn = x2.shape[1]
        dht, dxd = ops.highway_ht_bwd(dy.contiguous().view(-1, Cn), ht, x2)
        # dx = dht [W_H ; W_T] + dy (1 - T): the weight of that product is the (C, 2C) "Linear weight" [W_H^T | W_T^T]
        dx = ops.gemm(dht, ops.cat_params([Hw.detach(), Tw.detach()], transposed=True), res=dxd)
        dw, db = ops.gemm_wgrad(dht, x2, with_db=True, later=ops.grad_first(Hw, Tw, *ctx.hb), params=(Hw, Tw) + ctx.hb)
        return dx.view(*ctx.lead, Cn), dw[:Cn], db[:Cn], dw[Cn:], db[Cn:]


def highway_layer(x, Hw, Hb, Tw, Tb):
    return _HighwayLayerFn.apply(x, Hw, Hb, Tw, Tb)


class _GatherFn(Function):
    """rows of a table (F.embedding, src/embed.py:97-101) with scatter-add backward"""

    @staticmethod
    def forward(ctx, table, idx):
        ctx.save_for_backward(idx)
        ctx.V = table.shape[0]
        return ops.gather_rows(table, idx)

    @staticmethod
    def backward(ctx, dout):
        idx, = ctx.saved_tensors
        d = dout.contiguous()
        return ops.scatter_add_rows(d.view(-1, d.shape[-1]), idx.reshape(-1), ctx.V), None


def gather(table, idx):
    return _GatherFn.apply(table, idx)


class _SoftmaxArgmaxFn(Function):
    """p = softmax(logits, -1) with the argmax as a by-product (SeperateEmbedding.forward, src/embed.py:190-193)"""

    @staticmethod
    def forward(ctx, logits):
        p, idx = ops.softmax_argmax(logits.contiguous())
        ctx.save_for_backward(p)
        ctx.mark_non_differentiable(idx)
        return p, idx

    @staticmethod
    def backward(ctx, dp, _didx):
        p, = ctx.saved_tensors
        return ops.softmax_bwd(p, dp.contiguous())


def softmax_argmax(logits):
    return _SoftmaxArgmaxFn.apply(logits)


class _VqL2Fn(Function):
    """L2Embedding.forward (src/embed.py:105-147): p_code = softmax(relu(temp) * neg_batch_l2(x, table)), idx = argmax,
    new_latent = x + picked - x.detach() (straight-through).  Backward (what autograd derives there):
        g      = relu(temp) * softmax'(dp_code [+ dnew_latent E^T with the ST-onehot code])              (n, V)
        dx     = dnew_latent + 2 (g E) - 2 x rowsum(g)
        dtable = scatter_add(dnew_latent by idx) + [rows < n_real:] 2 (g^T x) - 2 E colsum(g)
        dtemp  = [temp > 0] sum(g * meas) / relu(temp)           (a learnable temperature, embed.py:34)
    `n_real` rows (first_n_real_mel * S, or all) let the p_code objectives reach the table (:115-122).
    stop_grad=True picks table[idx] (:134); False is the ST-onehot code p_hard @ table with p_hard = p + (onehot - p).detach()
    (:137-138): the same value up to one rounding of p + (1 - p), and its gradient also reaches p_code."""

    @staticmethod
    def forward(ctx, x, table, temp, n_real, st_onehot):
        x = x.contiguous()
        p, idx, out = ops.vq_l2(x, table, temp)
        ctx.save_for_backward(x, table, temp, p, idx)
        ctx.n_real, ctx.st_onehot = n_real, bool(st_onehot)
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)          # (the text-first cycle without unpaired text never uses the quantised latents: no zero gradient for them)
        return p, out, idx

    @staticmethod
    def backward(ctx, dp, dlat, _didx):
        x, table, temp, p, idx = ctx.saved_tensors
        V, D = table.shape
        if dp is None and dlat is None:
            return None, None, None, None, None
        x2 = _rows(x)
        dl2 = _rows(dlat.contiguous()) if dlat is not None else None
        n = x2.shape[0]
        n_real = n if ctx.n_real is None else ctx.n_real
        tab = table.detach().contiguous()
        dp2 = _rows(dp.contiguous()) if dp is not None else ops.zeros(n, V, device=x.device)
        if ctx.st_onehot and dl2 is not None:
            dp2 = dp2 + ops.gemm(dl2, tab)                                           # d p_hard = dnew_latent E^T   (n, V)
        g, rs = ops.softmax_bwd(_rows(p), dp2, 1.0, temp, want_rowsum=True)
        dx = dtab = dtemp = None
        if ctx.needs_input_grad[0]:
            ge = ops.gemm(g, tab.t().contiguous())                                   # (n, D) = g E
            dx = ops.rowscale_combine(ge, 2.0, x2, rs.reshape(-1), -2.0, dl2).view(x.shape)
        if ctx.needs_input_grad[1]:
            dtab = ops.scatter_add_rows(dl2, idx.reshape(-1), V) if dl2 is not None else None
            if n_real > 0:
                gx = ops.gemm_wgrad(g[:n_real], x2[:n_real])                       # (V, D) = g^T x over the real rows
                cs = ops.colsum(g[:n_real])
                dtab = ops.rowscale_combine(gx, 2.0, tab, cs, -2.0, dtab)
        if ctx.needs_input_grad[2]:
            # sim = relu(temp) * meas, meas = -(|x|^2 + |e|^2 - 2 x.e): d temp = sum(g / relu(temp) * meas) for temp > 0
            meas = 2.0 * ops.gemm(x2, tab) - x2.pow(2).sum(1, keepdim=True) - tab.pow(2).sum(1)
            t = temp.detach().clamp_min(0.0)
            dtemp = torch.where(t > 0, (g * meas).sum() / t.clamp_min(1e-30), torch.zeros_like(t)).view(temp.shape)
        return dx, dtab, dtemp, None, None


def vq_l2(x, table, temp, n_real=None, st_onehot=False):
    """-> (p_code, new_latent, idx)"""
    return _VqL2Fn.apply(x, table, temp, n_real, st_onehot)


class _StOnehotCodeFn(Function):
    """ST-onehot code of SeperateEmbedding (src/embed.py:198-203): p_hard @ table with p_hard = p + (onehot(idx) - p).detach() --
    the value of table[idx] (up to one rounding of p + (1 - p)); the gradient reaches p (d p = d code @ table^T) and the table rows
    (scatter-add, as p_hard is the one-hot matrix in value)."""

    @staticmethod
    def forward(ctx, p, table, idx):
        ctx.save_for_backward(table, idx)
        ctx.p_shape = p.shape
        return ops.gather_rows(table.detach().contiguous(), idx)

    @staticmethod
    def backward(ctx, dout):
        table, idx = ctx.saved_tensors
        d2 = _rows(dout.contiguous())
        dp = ops.gemm(d2, table.detach().contiguous()).view(ctx.p_shape) if ctx.needs_input_grad[0] else None
        dtab = ops.scatter_add_rows(d2, idx.reshape(-1), table.shape[0]) if ctx.needs_input_grad[1] else None
        return dp, dtab, None


def st_onehot_code(p, table, idx):
    return _StOnehotCodeFn.apply(p, table, idx)


class _BiLstmFn(Function):
    """Bidirectional nn.LSTM layer over precomputed input projections (lengths ignored, zero initial state).
    ref: src/module.py:432-438,:458-460"""

    @staticmethod
    def forward(ctx, xp_f, xp_b, w_hh_f, b_hh_f, w_hh_b, b_hh_b):
        B, T, H4 = xp_f.shape
        H = H4 // 4
        dev = xp_f.device
        out = torch.empty(B, T, 2 * H, device=dev, dtype=torch.float32)
        gs = [torch.empty(T, B, 4, H, device=dev, dtype=torch.float32) for _ in range(2)]
        cs = [torch.empty(T, B, H, device=dev, dtype=torch.float32) for _ in range(2)]
        ops.lstm_seq2(xp_f.contiguous(), xp_b.contiguous(), w_hh_f, w_hh_b, b_hh_f, b_hh_b, out, gs, cs)
        ctx.save_for_backward(out, w_hh_f, w_hh_b, gs[0], cs[0], gs[1], cs[1], b_hh_f, b_hh_b)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, w_hh_f, w_hh_b, g_f, c_f, g_b, c_b, b_hh_f, b_hh_b = ctx.saved_tensors
        dout = dout.contiguous()
        H = w_hh_f.shape[1]
        res = []
        if H % 16 == 0:     # packed operands: one launch per step for both directions (product + pointwise backward in its epilogue)
            dxps = ops.lstm_seq2_bwd(dout, (g_f, g_b), (c_f, c_b), None, (w_hh_f, w_hh_b))
        else:
            dxps = ops.lstm_seq2_bwd(dout, (g_f, g_b), (c_f, c_b), (ops.dx_weight(w_hh_f.detach())[0], ops.dx_weight(w_hh_b.detach())[0]))
        par = ((w_hh_f, b_hh_f), (w_hh_b, b_hh_b))
        for d in range(2):
            dxp = dxps[d]
            # h_{t-1} of the forward direction is out[t-1] (zero at t = 0): a 1-tap "conv" with pad +1 / -1
            # (the bias gradient = the column sums of dxp rides in the product's launches; the product itself may wait: ops.flush_wgrads)
            dw, db = ops.gemm_wgrad(dxp, out[:, :, d * H:(d + 1) * H], 1, 1 if d == 0 else -1, out=ops.grad_slot((w_hh_f, w_hh_b)[d]),
                                    with_db=True, db_out=ops.grad_slot((b_hh_f, b_hh_b)[d]), later=ops.grad_first(*par[d]), params=par[d])
            res.append((dxp, dw, db))
        return res[0][0], res[1][0], res[0][1], res[0][2], res[1][1], res[1][2]


def bilstm(xp_f, xp_b, w_hh_f, b_hh_f, w_hh_b, b_hh_b):
    return _BiLstmFn.apply(xp_f, xp_b, w_hh_f, b_hh_f, w_hh_b, b_hh_b)


class _LstmFn(Function):
    """Unidirectional nn.LSTM layer (forward in time) over precomputed input projections, zero initial state: the speech encoder
    with `rnn_bid: False` (src/asr.py:35-37).  One launch per time step (st_lstm_seq_fwd / st_lstm_seq_bwd)."""

    @staticmethod
    def forward(ctx, xp, w_hh, b_hh):
        B, T, H4 = xp.shape
        H = H4 // 4
        dev = xp.device
        out = torch.empty(B, T, H, device=dev, dtype=torch.float32)
        g = torch.empty(T, B, 4, H, device=dev, dtype=torch.float32)
        c = torch.empty(T, B, H, device=dev, dtype=torch.float32)
        ops.lstm_seq(xp.contiguous(), w_hh, b_hh, out, 0, False, gates_tape=g, c_tape=c)
        ctx.save_for_backward(out, w_hh, g, c, b_hh)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, w_hh, g, c, b_hh = ctx.saved_tensors
        dxp = ops.lstm_seq_bwd(dout.contiguous(), 0, g, c, ops.dx_weight(w_hh.detach())[0], False)
        # h_{t-1} is out[t-1] (zero at t = 0): a 1-tap "conv" with pad +1
        dw, db = ops.gemm_wgrad(dxp, out, 1, 1, out=ops.grad_slot(w_hh), with_db=True, db_out=ops.grad_slot(b_hh),
                                later=ops.grad_first(w_hh, b_hh), params=(w_hh, b_hh))
        return dxp, dw, db


def lstm(xp, w_hh, b_hh):
    return _LstmFn.apply(xp, w_hh, b_hh)


class _BiGruFn(Function):
    """Bidirectional nn.GRU layer over precomputed input projections.  ref: src/module.py:585-586,:617"""

    @staticmethod
    def forward(ctx, gi_f, gi_b, w_hh_f, b_hh_f, w_hh_b, b_hh_b):
        B, T, H3 = gi_f.shape
        H = H3 // 3
        out = torch.empty(B, T, 2 * H, device=gi_f.device, dtype=torch.float32)
        tape = torch.empty(2, B, T, 4, H, device=gi_f.device, dtype=torch.float32)
        ops.gru_seq(gi_f.contiguous(), gi_b.contiguous(), w_hh_f, w_hh_b, b_hh_f, b_hh_b, out, tape)
        ctx.save_for_backward(out, tape, w_hh_f, w_hh_b, b_hh_f, b_hh_b)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, tape, w_hh_f, w_hh_b, b_hh_f, b_hh_b = ctx.saved_tensors
        H = w_hh_f.shape[1]
        dgi_f, dgi_b, dgh_f, dgh_b = ops.gru_seq_bwd(dout.contiguous(), out, tape, w_hh_f, w_hh_b)
        par = (w_hh_f, b_hh_f, w_hh_b, b_hh_b)
        (dw_f, db_f), (dw_b, db_b) = ops.gemm_wgrad_batch([
            dict(dc=dgh_f, a=out[:, :, :H], KT=1, pad=1, out=ops.grad_slot(w_hh_f), with_db=True, db_out=ops.grad_slot(b_hh_f)),
            dict(dc=dgh_b, a=out[:, :, H:], KT=1, pad=-1, out=ops.grad_slot(w_hh_b), with_db=True, db_out=ops.grad_slot(b_hh_b))],
            later=ops.grad_first(*par), params=par)
        return dgi_f, dgi_b, dw_f, db_f, dw_b, db_b


def bigru(gi_f, gi_b, w_hh_f, b_hh_f, w_hh_b, b_hh_b):
    return _BiGruFn.apply(gi_f, gi_b, w_hh_f, b_hh_f, w_hh_b, b_hh_b)


class _CtcLossFn(Function):
    """torch.nn.CTCLoss()(log(prob + eps).transpose(0, 1), non-zero tokens of text, all frames, tokens per row): the paired ASR loss
    of bin/train_vqvae.py:430-444.  Value and gradient come from ONE kernel launch (st_ctc_loss); backward scales the stored
    gradient by the incoming scalar (like freq_loss)."""

    @staticmethod
    def forward(ctx, prob, text, eps, log_input=False):
        loss, dprob = ops.ctc_loss(prob.contiguous(), text.contiguous(), eps, want_grad=True, log_input=log_input)
        ctx.save_for_backward(dprob)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        dprob, = ctx.saved_tensors
        from . import _lib
        out = torch.empty_like(dprob)
        _lib.check(_lib.load().st_scale_by(ops._p(dprob), ops._p(dloss.contiguous()), ops._p(out), dprob.numel(), ops.stream_handle()),
                   'st_scale_by')
        return out, None, None, None


def ctc_loss(prob, text, eps=1e-10, apply_log=True):
    """prob (B, T, V) posteriors over the codebook (index 0 = blank), text (B, L) int64 (zeros = padding);
    apply_log=False: prob holds log-probabilities already (compute_ctcloss(..., apply_log=False), bin/train_vqvae.py:430-434)"""
    return _CtcLossFn.apply(prob, text, eps, not apply_log)


# --------------------------------------------------------------------------------------------- decoder loop
_untile_tape = ops.untile_tape


class _DecoderFn(Function):
    """The whole teacher-forced decode loop as one autograd node: forward = st_decoder_forward (the inference
    kernels, plus the gate tapes), backward = st_decoder_backward + batched TN GEMMs over the tapes.
    ref: src/module.py:184-288"""

    @staticmethod
    def forward(ctx, dec, plan, memory, pm, ada_std, ada_mean, teacher_pre, dec_in0, *params):
        mel, align, stop, tapes = dec._run_loop(plan, memory, pm, ada_std, ada_mean, teacher_pre, keep_tapes=True)
        ctx.dec, ctx.plan, ctx.tapes = dec, plan, tapes
        ctx.set_materialize_grads(False)          # (no zero tensors for outputs nobody differentiates: alignment and stop logits in training)
        ctx.has_in0 = dec_in0 is not None
        ctx.save_for_backward(memory, pm, ada_std, ada_mean, teacher_pre, align, mel, *params)
        return mel, align, stop

    @staticmethod
    def backward(ctx, dmel, dalign, dstop):
        import ctypes as C
        from . import _lib
        from ._lib import StDecoderBwdWeights, StDecoderBwdIO, StDecoderDims
        dec, plan, tapes = ctx.dec, ctx.plan, ctx.tapes
        memory, pm, ada_std, ada_mean, teacher_pre, align, mel_fwd = ctx.saved_tensors[:7]
        (pre_w0, pre_w1, q_w_ih, q_w_hh, q_b_ih, q_b_hh, wq, v, wc, wl, d_w_ih, d_w_hh, d_b_ih, d_b_hh,
         proj_w, proj_b, gate_w, gate_b) = ctx.saved_tensors[7:25]
        pre_norm = ctx.saved_tensors[25:]          # normalised prenet: (weight, bias) of the two norms
        steps, src, Bt = plan['steps'], plan['step_src'], plan['Bt']
        B, L, E = memory.shape
        r, n_mels, P = dec.n_frames_per_step, dec.n_mels, dec.prenet_dim
        Q, D, A, F, K = dec.query_rnn_dim, dec.dec_rnn_dim, dec.attn_dim, dec.n_location_filters, dec.location_kernel_size
        # own-output feedback (scheduled sampling steps / rows without a teacher): the gradient of dec_in_{t+1} flows into mel_t
        own = Bt != B or any(s == -1 for s in src[:steps - 1])
        dev = memory.device
        f32 = dict(device=dev, dtype=torch.float32)
        lib = _lib.load()
        kb = ops.kb16
        Bp = ((B + 15) // 16) * 16
        in_dim = r * n_mels
        XQw, XDw, XOw = P + E + Q, E + Q + D, D + E
        _, q_mask, d_mask = plan['masks']
        # forward tapes, un-tiled once: xq = [dec_in | ctx_{t-1} | h_q_{t-1}], xd = [ctx_t | adapted h_q_t | h_d_{t-1}], xo = [h_d_t | ctx_t]
        q_kbs, d_kbs, o_kbs = kb(P) + kb(E) + kb(Q), kb(E) + kb(Q) + kb(D), kb(D) + kb(E)
        XQ = _untile_tape(tapes['xq'], steps + 1, Bp, q_kbs, [(0, P), (kb(P), E), (kb(P) + kb(E), Q)])
        XD = _untile_tape(tapes['xd'], steps, Bp, d_kbs, [(0, E), (kb(E), Q), (kb(E) + kb(Q), D)])
        XO = tapes.get('xo_nat')
        if XO is None:
            XO = _untile_tape(tapes['xo'], steps, Bp, o_kbs, [(0, D), (kb(D), E)])
        hq_all = XQ[1:steps + 1].reshape(-1, XQw)[:, P + E:]               # h_q_t (dropped-out), rows (t, b)
        pq_all = ops.gemm(hq_all, wq)                                       # (steps*Bp, A)
        # output gradients through proj (+) gate for all steps at once
        # teacher forcing: the rows of dY are padded to a multiple of 4 floats (zero columns) -- 16-byte addressable rows put the two
        # products over them (dxo = dY [W_proj ; W_gate], the weight gradient dY^T [h_d | ctx]) on the LDS-DMA kernels (in_dim + 1 = 241
        # at r = 3: 59 + 50 us on the element-wise ones).  Own-output feedback keeps the exact width (the loop indexes dY itself).
        YW = in_dim + 1
        YWp = YW if own else (YW + 3) // 4 * 4
        # (the pack launch writes every element of the rows b < B, pad columns included; pad rows only exist when B is not a multiple of 16)
        dY = torch.empty(steps, Bp, YWp, **f32) if Bp == B else torch.zeros(steps, Bp, YWp, **f32)
        _lib.check(lib.st_decoder_pack_dout(ops._p(dmel.contiguous()) if dmel is not None else None,
                                            ops._p(dstop.contiguous()) if dstop is not None else None,
                                            ops._p(dY), YWp, B, Bp, steps, r, n_mels, ops.stream_handle()), 'st_decoder_pack_dout')
        dY2 = dY.view(-1, YWp)
        # [W_proj ; W_gate]^T (D+E, YWp), zero pad columns: a cached parameter layout, refreshed with all the others by one launch per step
        wpg_t = ops.cat_params([proj_w.detach(), gate_w.detach()], transposed=True, pad_to=YWp if YWp != YW else 0)
        # teacher forcing: no step's input depends on an earlier output, so all steps go through proj (+) gate at once;
        # with own-output feedback the loop forms dxo_t after adding the feedback gradient to dmel_t
        dxo = torch.zeros(steps * Bp, XOw, **f32) if own else ops.gemm(dY2, wpg_t)          # (steps*Bp, D+E)

        # every zero-initialised buffer of this backward comes out of ONE allocation and ONE fill launch
        # (the fused loop only where the library takes the dimensions: asked beforehand, because the step tapes below are handed over
        # uninitialised in that form -- st_decoder_backward refuses fuse_pw where it cannot honour it instead of falling back)
        dims = StDecoderDims(B=B, L=L, E=E, n_mels=n_mels, r=r, P=P, Q=Q, D=D, A=A, F=F, K=K, fuse_pre0=0)
        fuse_pw = (not own) and dec.bwd_fuse_pointwise and bool(lib.st_decoder_bwd_fuse_dims(C.byref(dims)))
        big = dict(dgq=(steps, Bp, 4 * Q), dgd=(steps, Bp, 4 * D), dxq=(steps + 1, Bp, XQw), dxd=(steps + 1, Bp, XDw), dpq=(steps, Bp, A))
        zshapes = dict(dcq=(B, Q), dcd=(B, D), dh0=(B, 2, L), dh1=(B, 2, L), dcum=(B, L), dhq_attn=(B, Q),
                       dgq_t16=(ops.t16_floats(B, 4 * Q),), dgd_t16=(ops.t16_floats(B, 4 * D),))
        # the step tapes of the fused loop are written in full by its launches when there are no pad rows (the step after the last one is
        # passed as absent, not as a zero slot): no 140 MB fill in front of the loop.  Every other form starts from zeros.
        lazy_big = fuse_pw and Bp == B and bool(getattr(dec, 'bwd_overlap_attn', False))
        if not lazy_big:
            zshapes.update(big)
        if fuse_pw:
            zshapes.update(dgd_t16_b=(ops.t16_floats(B, 4 * D),), dpq_t16=(ops.t16_floats(B, A),))
        numel = lambda shp: int(torch.Size(shp).numel())
        pool = ops.zeros(sum((numel(s) + 3) // 4 * 4 for s in zshapes.values()), **f32)      # (16-byte aligned pieces)
        zb, off = {}, 0
        for k, shp in zshapes.items():
            zb[k] = pool[off:off + numel(shp)].view(*shp)
            off += (numel(shp) + 3) // 4 * 4
        z = lambda *shape: ops.zeros(*shape, **f32)
        e_ = lambda *shape: ops.uninit(*shape, **f32)
        if lazy_big:
            zb.update({k: e_(*shp) for k, shp in big.items()})
        dgq, dgd, dxq, dxd, dpq = zb['dgq'], zb['dgd'], zb['dxq'], zb['dxd'], zb['dpq']
        ds_tape, loc_tape, dloc_tape = e_(steps, B, L, A), e_(steps, B, L, F), e_(steps, B, L, F)
        hist_tape, dctx_tape, dv_tape = e_(steps, B, L, 2), e_(steps, B, E), e_(steps, B, A)
        dcq, dcd, dh0, dh1, dcum, dhq_attn = (zb[k] for k in ('dcq', 'dcd', 'dh0', 'dh1', 'dcum', 'dhq_attn'))
        wt = dict(pq=ops.dx_weight(wq.detach())[0])                                         # W_q^T (Q, A): cached parameter layout
        bw = StDecoderBwdWeights()
        bw.attn_query_w_t = ops._p(wt['pq'])
        bw.attn_v, bw.attn_loc_conv_w, bw.attn_loc_lin_w = ops._p(v), ops._p(wc), ops._p(wl)
        # the two big per-step products dgates . W stream W^T = [W_ih | W_hh]^T in MFMA lane order: packed once per backward, straight
        # from the two parameters (no cat / transpose copies; the natural transposed forms are not needed next to the packed ones)
        wt['q_p16'] = ops.pack_weight_t([q_w_ih.detach(), q_w_hh.detach()])                 # N = P+E+Q, K = 4Q
        wt['d_p16'] = ops.pack_weight_t([d_w_ih.detach(), d_w_hh.detach()])                 # N = E+Q+D, K = 4D
        bw.q_w_cat_t_p16, bw.d_w_cat_t_p16 = ops._p(wt['q_p16']), ops._p(wt['d_p16'])
        dgq_t16, dgd_t16 = zb['dgq_t16'], zb['dgd_t16']
        io = StDecoderBwdIO()
        io.memory, io.pm, io.ada_std, io.align = ops._p(memory), ops._p(pm), ops._p(ada_std), ops._p(align)
        io.wcum_tape, io.cq_tape, io.cd_tape = ops._p(tapes['wcum']), ops._p(tapes['cq']), ops._p(tapes['cd'])
        io.gates_q_tape, io.gates_d_tape = ops._p(tapes['gates_q']), ops._p(tapes['gates_d'])
        io.q_mask, io.d_mask, io.pq_all = ops._p(q_mask), ops._p(d_mask), ops._p(pq_all)
        io.steps, io.Bp = steps, Bp
        io.dxo = ops._p(dxo)
        io.dalign = ops._p(dalign.contiguous()) if dalign is not None else None
        io.dgq, io.dgd, io.dxq, io.dxd, io.dpq = ops._p(dgq), ops._p(dgd), ops._p(dxq), ops._p(dxd), ops._p(dpq)
        if tapes.get('attn_loc') is not None and tapes.get('attn_s') is not None and tapes['attn_s'].dim() == 4:
            # the forward kept S_t = pm + W_l loc_t and loc_t of every step: the attention backward starts from them
            loc_tape = tapes['attn_loc']
            io.attn_s_tape = ops._p(tapes['attn_s'])
        io.ds_tape, io.loc_tape, io.dloc_tape = ops._p(ds_tape), ops._p(loc_tape), ops._p(dloc_tape)
        io.hist_tape, io.dctx_tape, io.dv_tape = ops._p(hist_tape), ops._p(dctx_tape), ops._p(dv_tape)
        io.dcq, io.dcd, io.dcum, io.dhq_attn = ops._p(dcq), ops._p(dcd), ops._p(dcum), ops._p(dhq_attn)
        io.dhist[0], io.dhist[1] = ops._p(dh0), ops._p(dh1)
        io.dgq_t16, io.dgd_t16 = ops._p(dgq_t16), ops._p(dgd_t16)
        if fuse_pw:      # pointwise LSTM backward in the epilogues of the loop's products (4 launches per step instead of 6)
            wt['pq_p16'] = ops.pack_weight([wt['pq']], [A], Q)
            bw.attn_query_w_t_p16 = ops._p(wt['pq_p16'])
            io.fuse_pw, io.dgd_t16_b, io.dpq_t16 = 1, ops._p(zb['dgd_t16_b']), ops._p(zb['dpq_t16'])
            io.overlap_attn = 1 if dec.bwd_overlap_attn else 0
            parts = int(getattr(dec, 'bwd_attn_parts', 2))
            if io.overlap_attn and parts > 1:      # the hosted attention backward as `parts` workgroups per utterance (decoder_bwd.hip)
                dloc_part = e_(parts, B, L, F)
                io.attn_parts, io.dloc_part = parts, ops._p(dloc_part)
                dsplits = int(getattr(dec, 'bwd_dxd_splits', 2))
                if parts == 2 and dsplits > 0 and 16 < B <= 32 and Bp == B:      # the hosted launch's product K-split into slabs
                    dxd_part = e_(dsplits, B, XDw)
                    io.dxd_part, io.dxd_splits = ops._p(dxd_part), dsplits
                    qsplits = int(getattr(dec, 'bwd_dxq_splits', 4))
                    if lazy_big and 0 < qsplits <= 4:                            # ... and the query cell's (its consumers add the slabs up)
                        dxq_part = e_(steps + 1, qsplits, B, XQw)
                        io.dxq_part, io.dxq_splits = ops._p(dxq_part), qsplits
        src_arr = (C.c_int * max(steps, 1))(*src)
        io.step_src, io.Bt = C.cast(src_arr, C.POINTER(C.c_int)), Bt
        io.need_dxq0 = 1 if ctx.has_in0 else 0
        if own:
            own_mask = plan['masks'][0]
            pre1_nat = _untile_tape(tapes['pre1'], steps, Bp, kb(P), [(0, P)])
            d2_tape, dp1_tape = z(steps, Bp, P), z(steps, Bp, P)
            wt.update(w1=pre_w1.detach().t().contiguous(), w0=pre_w0.detach().t().contiguous())
            tmp_p, tmp_in = z(B, P), z(B, in_dim)
            io.dY, io.dxo_rw, io.wpg_t = ops._p(dY), ops._p(dxo), ops._p(wpg_t)
            io.pre_w1_t, io.pre_w0_t, io.own_mask = ops._p(wt['w1']), ops._p(wt['w0']), ops._p(own_mask)
            io.xq_nat, io.pre1_nat = ops._p(XQ), ops._p(pre1_nat)
            io.d2_tape, io.dp1_tape, io.tmp_p, io.tmp_in = ops._p(d2_tape), ops._p(dp1_tape), ops._p(tmp_p), ops._p(tmp_in)
            if pre_norm:
                # through LayerNorm / BatchNorm1d of the two prenet layers: d2_tape / dp1_tape end up holding the gradients at the Linear
                # outputs (what dW1 / dW0 need); the norms' own gradients are summed over the steps by the loop
                dnorm = z(4, P)                                         # d weight_0, d bias_0, d weight_1, d bias_1
                io.prenet_norm, io.pre_y_tape = int(tapes['pn_mode']), ops._p(tapes['pre_y'])
                for l, nm in enumerate(lyr.norm for lyr in dec.prenet.layers):
                    io.pre_norm_w[l] = ops._p(pre_norm[2 * l])
                    if tapes['pn_mode'] == 2:
                        io.pre_norm_rm[l], io.pre_norm_rv[l] = ops._p(nm.running_mean), ops._p(nm.running_var)
                    io.dpre_norm_w[l], io.dpre_norm_b[l] = ops._p(dnorm[2 * l]), ops._p(dnorm[2 * l + 1])
                io.pre_norm_eps = float(dec.prenet.layers[0].norm.eps)
        # (does the loop leave the gradient w.r.t. the query cell's inputs as K-split slabs?  The library decides by shape; ask it)
        dxq_slabs = bool(int(lib.st_decoder_bwd_forms(C.byref(dims), C.byref(io))) & 4)
        _lib.check(lib.st_decoder_backward(C.byref(bw), C.byref(dims), C.byref(io), ops.stream_handle()), 'st_decoder_backward')

        # weight gradients: TN GEMMs over the tapes (rows = (step, utterance); pad rows are zero)
        dgq2, dgd2 = dgq.view(-1, 4 * Q), dgd.view(-1, 4 * D)
        # the gradients of [W_ih | W_hh] leave the product's slab sum as the two parameters' own tensors, the bias gradient twice (b_ih
        # and b_hh receive the same sums): no slicing copies -- and under data parallelism all of them are written straight into
        # their all-reduce bucket slots (ops.grad_slot)
        dwq_ih, dwq_hh, dbq, dbq2 = ops.gemm_wgrad_split(dgq2, XQ[:steps].reshape(-1, XQw), P + E, q_w_ih, q_w_hh, q_b_ih, q_b_hh, with_db=True)
        dwd_ih, dwd_hh, dbd, dbd2 = ops.gemm_wgrad_split(dgd2, XD.view(-1, XDw), E + Q, d_w_ih, d_w_hh, d_b_ih, d_b_hh, with_db=True)
        # three more products over the tapes in one launch (+ one slab sum): proj (+) gate, the query projection, W_l (sums over the steps of
        # the per-step tape slices)
        tail_par = (proj_w, proj_b, gate_w, gate_b, wq, wl)
        (dwpg, dbpg), (dwq_attn, _), (dwl, _) = ops.gemm_wgrad_batch([
            dict(dc=dY2, a=XO.view(-1, XOw), with_db=True),
            dict(dc=dpq.view(-1, A), a=hq_all, out=ops.grad_slot(wq)),
            dict(dc=ds_tape.view(-1, A), a=loc_tape.view(-1, F))],                              # (A, F)
            later=ops.grad_first(*tail_par), params=tail_par)
        dv = ops.colsum(dv_tape.view(-1, A)).view(v.shape)
        dwc = ops.gemm_wgrad(dloc_tape.view(steps * B, L, F), hist_tape.view(steps * B, L, 2), K, (K - 1) // 2,
                             later=ops.grad_first(wc), params=(wc,))   # (F, 2, K)
        dpm = ops.colsum(ds_tape.view(steps, -1)).view(B, L, A)
        dmem = torch.empty(B, L, E, **f32)
        _lib.check(lib.st_attn_dmem(ops._p(align), ops._p(dctx_tape), ops._p(dmem), B, steps, L, E, ops.stream_handle()),
                   'st_attn_dmem')
        dstd, dmean = torch.empty(B, Q, **f32), torch.empty(B, Q, **f32)
        _lib.check(lib.st_adain_bwd(ops._p(dxd) + 4 * E, Bp * XDw, XDw, ops._p(XQ) + 4 * (Bp * XQw + P + E), Bp * XQw, XQw,
                                    ops._p(ada_std), ops._p(ada_mean), ops._p(dstd), ops._p(dmean), B, Q, steps,
                                    ops.stream_handle()), 'st_adain_bwd')
        # prenet weights on the own-output path: dW1 = d2^T pre1, dW0 = dp1^T mel over the steps / rows that fed back
        dpre_w0 = dpre_w1 = None
        if own:
            ops.grad_first(pre_w0, pre_w1)     # (announced: the teacher path's products of the same weights must not be queued behind these)
            mel_tb = z(steps, Bp, in_dim)                       # mel_t in (step, utterance) row order
            ops.copy3d(mel_tb.permute(1, 0, 2)[:B], mel_fwd.view(B, steps, in_dim), B, steps, in_dim)
            dpre_w1 = ops.gemm_wgrad(d2_tape.view(-1, P), pre1_nat.view(-1, P))
            dpre_w0 = ops.gemm_wgrad(dp1_tape.view(-1, P), mel_tb.view(-1, in_dim))
        # d prenet(teacher): step t+1 read teacher frame src[t]
        dteacher = None
        if teacher_pre is not None:
            Tt = teacher_pre.shape[1]
            identity = list(src[:steps - 1]) == list(range(steps - 1))
            if steps > 1 and identity and dxq_slabs and dxq_part.is_contiguous():
                # (plain teacher forcing: the slabs of dxq_t[:, :P] added in slab order, frames no step read zeroed -- one launch)
                dteacher = torch.empty(Bt, Tt, P, **f32)
                _lib.check(lib.st_decoder_dteacher_sum(ops._p(dxq_part), ops._p(dteacher), int(dxq_part.shape[1]), int(dxq_part.shape[2]),
                                                       int(dxq_part.shape[3]), Bt, Tt, P, steps, ops.stream_handle()), 'st_decoder_dteacher_sum')
                steps_done = True
            else:
                dteacher = z(Bt, Tt, P)
                steps_done = False
            if steps > 1 and not steps_done:
                if dxq_slabs and not identity:     # (rare: teacher-mean steps) the general forms below read whole rows: add the slabs up once
                    dxq[1:steps] = dxq_part[1:steps].sum(1)
                if identity and dxq_slabs:
                    for sl in range(qsplits):      # the slabs of dxq_t[:, :P], added in slab order
                        ops.copy3d(dteacher.view(Bt, Tt, P)[:, :steps - 1], dxq_part[1:steps, sl].permute(1, 0, 2)[:Bt, :, :P], Bt, steps - 1, P,
                                   accumulate=sl > 0)
                elif identity:
                    ops.copy3d(dteacher.view(Bt, Tt, P)[:, :steps - 1], dxq[1:steps].permute(1, 0, 2)[:Bt, :, :P], Bt, steps - 1, P)
                else:
                    for t in range(steps - 1):
                        if src[t] >= 0:
                            ops.copy3d(dteacher[:, src[t]:src[t] + 1], dxq[t + 1:t + 2].permute(1, 0, 2)[:Bt, :, :P], Bt, 1, P,
                                       accumulate=True)
                    # steps fed with teacher.mean(dim=1) (drop_dec_in, src/module.py:193-194): d mean / d frame = 1 / Tt for every frame
                    mean_steps = [t + 1 for t in range(steps - 1) if src[t] == -2]
                    if mean_steps:
                        dmean_in = dxq[mean_steps][:, :Bt, :P].sum(0) / Tt                 # (Bt, P)
                        dteacher += dmean_in.unsqueeze(1)
        c = lambda t: t.contiguous()
        ddec_in0 = dxq[0, :B, :P].contiguous() if ctx.has_in0 else None     # gradient of dec_in_0 = prenet(go frame)
        grads = (None, None, dmem, dpm, dstd, dmean, dteacher, ddec_in0,
                 dpre_w0, dpre_w1,                                                # prenet weights: only via own-output feedback
                 dwq_ih, dwq_hh, dbq, dbq2,
                 dwq_attn, dv, dwc, dwl,
                 dwd_ih, dwd_hh, dbd, dbd2,
                 c(dwpg[:in_dim]), c(dbpg[:in_dim]), c(dwpg[in_dim:in_dim + 1]), c(dbpg[in_dim:in_dim + 1]))
        if pre_norm:
            grads += tuple(dnorm[i] for i in range(4)) if own else (None,) * 4
        ctx.tapes = None
        return grads


def decoder_loop(dec, plan, memory, pm, ada_std, ada_mean, teacher_pre, dec_in0=None):
    params = (dec.prenet.layers[0].linear.weight, dec.prenet.layers[1].linear.weight,
              dec.query_rnn.weight_ih, dec.query_rnn.weight_hh, dec.query_rnn.bias_ih, dec.query_rnn.bias_hh,
              dec.attn.query_layer.linear.weight, dec.attn.v.linear.weight, *dec.attn.location_weights(),
              dec.dec_rnn.weight_ih, dec.dec_rnn.weight_hh, dec.dec_rnn.bias_ih, dec.dec_rnn.bias_hh,
              dec.proj.linear.weight, dec.proj.linear.bias, dec.gate_layer.linear.weight, dec.gate_layer.linear.bias)
    if dec.prenet_norm_type is not None:
        for lyr in dec.prenet.layers:
            params += (lyr.norm.weight, lyr.norm.bias)
    return _DecoderFn.apply(dec, plan, memory, pm, ada_std, ada_mean, teacher_pre, dec_in0, *params)


# --------------------------------------------------------------------------------------------- trainer loss
class _FreqLossFn(Function):
    @staticmethod
    def forward(ctx, pred, label, n_low, w_all, w_low, w_diff, l1):
        from . import _lib
        pred, label = _contig3(pred), _contig3(label)
        B, T, D = pred.shape
        loss = torch.empty((), device=pred.device, dtype=torch.float32)
        dpred = torch.empty_like(pred)
        ws = torch.empty(int(_lib.load().st_freq_loss_workspace_floats()), device=pred.device, dtype=torch.float32)
        _lib.check(_lib.load().st_freq_loss(ops._p(pred), ops._p(label), ops._p(loss), ops._p(dpred), ops._p(ws), B, T, D,
                                            int(n_low), float(w_all), float(w_low), float(w_diff), 1 if l1 else 0,
                                            ops.stream_handle()), 'st_freq_loss')
        ctx.save_for_backward(dpred)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        from . import _lib
        dpred, = ctx.saved_tensors
        out = torch.empty_like(dpred)
        _lib.check(_lib.load().st_scale_by(ops._p(dpred), ops._p(dloss.contiguous()), ops._p(out), dpred.numel(),
                                           ops.stream_handle()), 'st_scale_by')
        return out, None, None, None, None, None, None


def freq_loss(pred, label, sample_rate, n_mels, loss='mse', differential_loss=True, emphasize_linear_low=True):
    """the trainer's spectrogram loss (same signature as the reference, src/util.py:80-126) on the HIP path"""
    if loss not in ('l1', 'mse'):
        raise NotImplementedError(loss)
    dim = pred.shape[-1]
    w_all, w_low, w_diff, n_low = 1.0, 0.0, 0.0, 0
    if dim != n_mels and emphasize_linear_low:
        n_low = int(dim * (3000 / (sample_rate / 2)))
        w_all, w_low = 0.5, 0.5
    if dim == n_mels and differential_loss:
        w_diff = 0.5
    return _FreqLossFn.apply(pred, label, n_low, w_all, w_low, w_diff, loss == 'l1')


# --------------------------------------------------------------------------------------------- VQ run-length merge
class _MeanForwardFn(Function):
    """VQVAE.mean_forward (src/vqvae.py:218-257) on device; returns the worst-case padded (B, T, D) tensor + lens"""

    @staticmethod
    def forward(ctx, p_code, latent, max_frames):
        from . import _lib
        p_code, latent = p_code.contiguous(), latent.contiguous()
        B, T, D = latent.shape
        V = p_code.shape[-1]
        dev = latent.device
        out = torch.empty(B, T, D, device=dev, dtype=torch.float32)
        ops.fill_(out, 0.0)            # (a library launch, not torch.zeros: it is then part of a captured graph as well)
        lens = torch.empty(B, device=dev, dtype=torch.int32)
        seg = torch.empty(B, T, device=dev, dtype=torch.int32)
        w = torch.empty(B, T, device=dev, dtype=torch.float32)
        _lib.check(_lib.load().st_vq_mean_fwd(ops._p(p_code), ops._p(latent), ops._p(out), ops._p(lens, torch.int32),
                                              ops._p(seg, torch.int32), ops._p(w), B, T, D, V, int(max_frames),
                                              ops.stream_handle()), 'st_vq_mean_fwd')
        ctx.save_for_backward(seg, w)
        ctx.mark_non_differentiable(lens)
        return out, lens

    @staticmethod
    def backward(ctx, dout, _dlens):
        from . import _lib
        seg, w = ctx.saved_tensors
        dout = dout.contiguous()
        B, T = seg.shape
        D = dout.shape[-1]
        dlat = torch.empty(B, T, D, device=dout.device, dtype=torch.float32)
        _lib.check(_lib.load().st_vq_mean_bwd(ops._p(dout), dout.shape[1], ops._p(seg, torch.int32), ops._p(w), ops._p(dlat),
                                              B, T, D, ops.stream_handle()), 'st_vq_mean_bwd')
        return None, dlat, None


def mean_forward(p_code, latent, max_frames_per_phn):
    """-> (batch_latent (B, Nmax, D), lengths (B,) int64) or None when an utterance is all blank; ONE device->host copy
    (the lengths, which fix the output shape) instead of the reference's per-utterance `.cpu().tolist()` + host loop"""
    out, lens = _MeanForwardFn.apply(p_code, latent, max_frames_per_phn)
    lens_h = lens.cpu()
    if int(lens_h.min()) == 0:
        return None
    return out[:, :int(lens_h.max())], lens.long()
