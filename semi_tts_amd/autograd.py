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


class _ConvFn(Function):
    """y = [act(conv1d/linear(pool?(x)) + b) (+ res)] (* mask), channels-last"""

    @staticmethod
    def forward(ctx, x, w, b, res, mask, pad, Tout, act, pool_prev):
        assert not (res is not None and (act is not None or mask is not None)), 'residual only after a linear map'
        assert mask is None or act in (None, 'relu'), 'dropout mask is fused only after relu / identity'
        x = x.contiguous()
        y = ops.gemm(x, w, pad=pad, Tout=Tout, bias=b, act_pre=act, res=res, mask=mask, pool_prev=pool_prev)
        ctx.save_for_backward(x, w, y if act is not None else None, mask)
        ctx.cfg = (pad, act, pool_prev, b is not None, res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, mask = ctx.saved_tensors
        pad, act, pool_prev, has_b, has_res = ctx.cfg
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
        if ctx.needs_input_grad[0]:
            wt = (w.detach().permute(1, 0, 2).flip(2) if w.dim() == 3 else w.detach().t()).contiguous()
            dx = ops.gemm(dpre, wt, Bn=Bn, Tin=To, Tout=Tin, pad=KT - 1 - pad)
            if pool_prev:
                dx = ops.pool_prev_bwd(dx, x)
        if ctx.needs_input_grad[1]:
            dw = ops.gemm_wgrad(dpre, x, KT, pad, Bn=Bn, Tin=Tin, Tout=To, N=N, pool_prev=pool_prev).view(w.shape)
        if has_b and ctx.needs_input_grad[2]:
            db = ops.colsum(_rows(dpre))
        dres = dy if has_res and ctx.needs_input_grad[3] else None
        return dx, dw, db, dres, None, None, None, None, None


def conv(x, w, b=None, *, pad=0, Tout=None, act=None, res=None, mask=None, pool_prev=False):
    """differentiable ops.gemm (conv1d over channels-last rows, or linear when w is 2-D)"""
    return _ConvFn.apply(x, w, b, res, mask, pad, Tout, act, pool_prev)


def linear(x, w, b=None, act=None, mask=None):
    lead = x.shape[:-1]
    y = _ConvFn.apply(x.reshape(-1, x.shape[-1]), w, b, None,
                      mask.reshape(-1, mask.shape[-1]) if mask is not None else None, 0, None, act, False)
    return y.view(*lead, -1)


class _BnTrainFn(Function):
    """y = act(BatchNorm1d(x)) with batch statistics over all leading dims (training mode);
    updates running_mean / running_var in place like nn.BatchNorm1d (unbiased running var)."""

    @staticmethod
    def forward(ctx, x, weight, bias, run_mean, run_var, eps, momentum, act):
        x = x.contiguous()
        N = x.shape[-1]
        x2 = _rows(x)
        mean, var = ops.bn_stats(x2, 0, N, run_mean, run_var, momentum)
        y = ops.bn_norm(x2, 0, N, mean, var, weight, bias, eps, act).view(x.shape)
        ctx.save_for_backward(x, y if act is not None else None, mean, var, weight)
        ctx.cfg = (eps, act)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, var, weight = ctx.saved_tensors
        eps, act = ctx.cfg
        dx, dw, db = ops.bn_bwd(_rows(dy.contiguous()), _rows(y) if y is not None else None, act, _rows(x), mean, var,
                                weight, eps)
        return dx.view(x.shape), dw, db, None, None, None, None, None


def batch_norm_train(x, bn, act=None):
    y = _BnTrainFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, act)
    bn.num_batches_tracked += 1
    return y


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


class _BiLstmFn(Function):
    """Bidirectional nn.LSTM layer over precomputed input projections (lengths ignored, zero initial state).
    ref: src/module.py:432-438,:458-460"""

    @staticmethod
    def forward(ctx, xp_f, xp_b, w_hh_f, b_hh_f, w_hh_b, b_hh_b):
        B, T, H4 = xp_f.shape
        H = H4 // 4
        dev = xp_f.device
        out = torch.empty(B, T, 2 * H, device=dev, dtype=torch.float32)
        ws = torch.empty(3 * B * H, device=dev, dtype=torch.float32)
        tapes = []
        for d, (xp, w, b) in enumerate(((xp_f, w_hh_f, b_hh_f), (xp_b, w_hh_b, b_hh_b))):
            g = torch.empty(T, B, 4, H, device=dev, dtype=torch.float32)
            c = torch.empty(T, B, H, device=dev, dtype=torch.float32)
            ops.lstm_seq(xp.contiguous(), w, b, out, d * H, d == 1, ws, g, c)
            tapes += [g, c]
        ctx.save_for_backward(out, w_hh_f, w_hh_b, *tapes)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, w_hh_f, w_hh_b, g_f, c_f, g_b, c_b = ctx.saved_tensors
        dout = dout.contiguous()
        H = w_hh_f.shape[1]
        res = []
        for d, (w, g, c) in enumerate(((w_hh_f, g_f, c_f), (w_hh_b, g_b, c_b))):
            dxp = ops.lstm_seq_bwd(dout, d * H, g, c, w.detach().t().contiguous(), d == 1)
            # h_{t-1} of the forward direction is out[t-1] (zero at t = 0): a 1-tap "conv" with pad +1 / -1
            dw = ops.gemm_wgrad(dxp, out[:, :, d * H:(d + 1) * H], 1, 1 if d == 0 else -1)
            db = ops.colsum(_rows(dxp))
            res.append((dxp, dw, db))
        return res[0][0], res[1][0], res[0][1], res[0][2], res[1][1], res[1][2]


def bilstm(xp_f, xp_b, w_hh_f, b_hh_f, w_hh_b, b_hh_b):
    return _BiLstmFn.apply(xp_f, xp_b, w_hh_f, b_hh_f, w_hh_b, b_hh_b)


class _BiGruFn(Function):
    """Bidirectional nn.GRU layer over precomputed input projections.  ref: src/module.py:585-586,:617"""

    @staticmethod
    def forward(ctx, gi_f, gi_b, w_hh_f, b_hh_f, w_hh_b, b_hh_b):
        B, T, H3 = gi_f.shape
        H = H3 // 3
        out = torch.empty(B, T, 2 * H, device=gi_f.device, dtype=torch.float32)
        tape = torch.empty(2, B, T, 4, H, device=gi_f.device, dtype=torch.float32)
        ops.gru_seq(gi_f.contiguous(), gi_b.contiguous(), w_hh_f, w_hh_b, b_hh_f, b_hh_b, out, tape)
        ctx.save_for_backward(out, tape, w_hh_f, w_hh_b)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, tape, w_hh_f, w_hh_b = ctx.saved_tensors
        H = w_hh_f.shape[1]
        dgi_f, dgi_b, dgh_f, dgh_b = ops.gru_seq_bwd(dout.contiguous(), out, tape, w_hh_f, w_hh_b)
        dw_f = ops.gemm_wgrad(dgh_f, out[:, :, :H], 1, 1)
        dw_b = ops.gemm_wgrad(dgh_b, out[:, :, H:], 1, -1)
        return dgi_f, dgi_b, dw_f, ops.colsum(_rows(dgh_f)), dw_b, ops.colsum(_rows(dgh_b))


def bigru(gi_f, gi_b, w_hh_f, b_hh_f, w_hh_b, b_hh_b):
    return _BiGruFn.apply(gi_f, gi_b, w_hh_f, b_hh_f, w_hh_b, b_hh_b)
