"""Gradient parity of the HIP backward path (through the C ABI) against CPU autograd through the
oracle / the torch ops the reference's modules are made of.  Needs a real MI355X: pytest -m gpu

Tolerances are written next to each check; gradients are sums of O(rows) fp32 products, compared
with a float64 CPU evaluation and normalised by the gradient's own scale.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from helpers import maxdiff, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'the gpu-marked tests need a GPU'
    from semi_tts_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def relerr(a, b):
    b = b.detach().cpu().double()
    return float((a.detach().cpu().double() - b).abs().max() / (b.abs().max() + 1e-12))


def conv_cl_ref(x, w, b, pad, Tout, act, pool_prev=False, res=None, mask=None):
    """float64 torch restatement of the conv the reference builds from nn.Conv1d (+MaxPool1d(2,1,1)[:T])"""
    xt = x.transpose(1, 2)
    if pool_prev:
        xt = F.max_pool1d(xt, 2, 1, 1)[:, :, :x.shape[1]]
    y = F.conv1d(xt, w, b, padding=pad)[:, :, :Tout].transpose(1, 2)
    if act == 'relu':
        y = torch.relu(y)
    elif act == 'tanh':
        y = torch.tanh(y)
    elif act == 'sigmoid':
        y = torch.sigmoid(y)
    if res is not None:
        y = y + res
    if mask is not None:
        y = y * mask
    return y


@pytest.mark.parametrize('Bn,T,Cin,N,KT,pad', [
    (85 * 4, 43, 2, 32, 31, 15),      # the location conv over the step tapes (C2 shape, fewer sequences)
    (40, 171, 2, 32, 31, 15),         # C5's text length: three 64-row blocks per sequence
    (700, 9, 1, 8, 5, 2),             # one channel, short sequences: more workgroups than the slab count covers in one pass
    (3, 20, 4, 64, 16, 8),            # even kernel: Tout = T + 1
    (2, 130, 3, 12, 7, 0),            # no padding: Tout = T - 6
])
def test_small_channel_conv_weight_gradient(dev, Bn, T, Cin, N, KT, pad):
    """dW of a convolution with very few input channels (convw_small_kernel: im2col columns out of LDS) + the many-slab sum
    (sum_partials_tall_kernel) against float64 autograd.  ref: backward of the location conv, src/module.py:239,371-407"""
    from semi_tts_amd import ops
    Tout = T + 2 * pad - KT + 1
    x, dy = rnd(Bn, T, Cin, seed=5), rnd(Bn, Tout, N, seed=6)
    dw = ops.gemm_wgrad(dy.to(dev), x.to(dev), KT, pad, Tout=Tout)
    dw2 = ops.gemm_wgrad(dy.to(dev), x.to(dev), KT, pad, Tout=Tout)
    assert torch.equal(dw, dw2)
    w = torch.zeros(N, Cin, KT, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv1d(x.double().transpose(1, 2), w, padding=pad).transpose(1, 2)
    assert y.shape[1] == Tout
    y.backward(dy.double())
    e = relerr(dw, w.grad)
    report('small_channel_conv_wgrad', Bn=Bn, T=T, Cin=Cin, N=N, KT=KT, err=e)
    assert e < 2e-5


def test_attention_memory_gradient_and_wide_column_sums(dev):
    """st_attn_dmem (dmem = sum_t w_t (x) dctx_t, four positions x 256 dims per workgroup) and the few-rows / many-columns form of
    st_colsum (the processed-memory gradient: a sum over the step tapes) against float64"""
    from semi_tts_amd import ops, _lib
    lib = _lib.load()
    for B, steps, L, E in ((32, 85, 43, 512), (3, 7, 10, 12), (5, 4, 9, 260), (2, 3, 5, 6)):
        align, dctx = rnd(B, steps, L, seed=1), rnd(steps, B, E, seed=2)
        dmem = torch.empty(B, L, E, device=dev)
        ad, dd = align.to(dev), dctx.to(dev)
        _lib.check(lib.st_attn_dmem(ops._p(ad), ops._p(dd), ops._p(dmem), B, steps, L, E, ops.stream_handle()), 'st_attn_dmem')
        ref = torch.einsum('btl,tbe->ble', align.double(), dctx.double())
        assert relerr(dmem, ref) < 2e-6, (B, steps, L, E)
    for M, N in ((85, 32 * 43 * 256), (3, 65536), (200, 70000)):
        x = rnd(M, N, seed=3)
        s = ops.colsum(x.to(dev))
        assert relerr(s, x.double().sum(0)) < 2e-6, (M, N)


def test_queued_weight_gradients(dev):
    """Weight-gradient products are queued during a backward pass and leave together (ops.flush_wgrads).  A layer applied TWICE gives
    its weight two contributions: the second must find the first one written (it flushes the queue and runs at once) -- and a chain
    of different layers (each queued) must give the gradients of the unqueued run (ST_WGRAD_DEFER=0 semantics: later=False), bit for bit."""
    from semi_tts_amd import autograd as AG, ops
    M, C = 512, 64
    x1, x2 = rnd(M, C, seed=1), rnd(M, C, seed=2)
    ws = [rnd(C, C, scale=C ** -0.5, seed=10 + i) for i in range(5)]
    bs = [rnd(C, seed=20 + i) for i in range(5)]
    dy = rnd(M, C, seed=3)

    def run(defer):
        old = ops.WGRAD_DEFER
        ops.WGRAD_DEFER = defer
        try:
            wd = [w.to(dev).requires_grad_() for w in ws]
            bd = [b.to(dev).requires_grad_() for b in bs]
            a, b2 = x1.to(dev), x2.to(dev)
            for i in range(5):
                a = AG.linear(a, wd[i], bd[i], act='relu')
            b2 = AG.linear(b2, wd[0], bd[0], act='relu')          # layer 0 again, on another input
            (a + AG.linear(b2, wd[4], bd[4])).backward(dy.to(dev))   # ... and layer 4
            assert not ops._WQ, 'the queue must be empty when backward() returns'
            return [t.grad.clone() for t in wd + bd]
        finally:
            ops.WGRAD_DEFER = old

    g1, g0 = run(True), run(False)
    for a, b in zip(g1, g0):
        assert torch.equal(a, b)
    # against float64
    wr = [w.double().requires_grad_() for w in ws]
    br = [b.double().requires_grad_() for b in bs]
    a, b2 = x1.double(), x2.double()
    for i in range(5):
        a = torch.relu(a @ wr[i].t() + br[i])
    b2 = torch.relu(b2 @ wr[0].t() + br[0])
    (a + b2 @ wr[4].t() + br[4]).backward(dy.double())
    worst = max(relerr(g, r.grad) for g, r in zip(g1, wr + br))
    report('queued_weight_gradients', worst=worst)
    assert worst < 2e-5


def test_queued_weight_gradients_with_accumulation_and_hooks(dev):
    """A parameter that already holds a gradient (two backward() calls without zero_grad, or zero_grad(set_to_none=False)) makes
    AccumulateGrad add the new gradient the moment the backward function returns, and a tensor hook reads it there as well -- both before
    a queued product would have written it.  Such products must run at once (ops.grad_first): same gradients as ST_WGRAD_DEFER=0, bit for bit."""
    from semi_tts_amd import autograd as AG, ops
    M, C = 256, 32
    x1, x2 = rnd(M, C, seed=1), rnd(M, C, seed=2)
    ws = [rnd(C, C, scale=C ** -0.5, seed=30 + i) for i in range(3)]
    bs = [rnd(C, seed=40 + i) for i in range(3)]
    dy = rnd(M, C, seed=3)

    def run(defer):
        old = ops.WGRAD_DEFER
        ops.WGRAD_DEFER = defer
        seen = []
        try:
            wd = [w.to(dev).requires_grad_() for w in ws]
            bd = [b.to(dev).requires_grad_() for b in bs]
            wd[1].register_hook(lambda g: seen.append(g.detach().clone()))        # a hook reads the gradient as it is handed over
            for x in (x1, x2):                                                    # second pass: every .grad exists already
                a = x.to(dev)
                for i in range(3):
                    a = AG.linear(a, wd[i], bd[i], act='relu')
                a.backward(dy.to(dev))
                assert not ops._WQ
            torch.cuda.synchronize()
            wd[0].grad.zero_(); bd[0].grad.zero_()                               # zero_grad(set_to_none=False) on one layer, then a third pass
            a = x1.to(dev)
            for i in range(3):
                a = AG.linear(a, wd[i], bd[i], act='relu')
            a.backward(dy.to(dev))
            return [t.grad.clone() for t in wd + bd] + seen
        finally:
            ops.WGRAD_DEFER = old

    g1, g0 = run(True), run(False)
    assert len(g1) == len(g0) == 6 + 3
    for a, b in zip(g1, g0):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b)


@pytest.mark.parametrize('B,T,Cin,N,KT,pad,act,pool', [
    (3, 37, 24, 40, 5, 2, None, False),        # encoder conv
    (2, 50, 80, 80, 4, 2, 'relu', False),      # even-k bank conv (Tout = T+1 and T below)
    (2, 33, 70, 65, 3, 1, 'relu', True),       # first projection conv with the fused max-pool
    (4, 20, 16, 130, 1, 0, 'sigmoid', False),  # k=1
    (2, 29, 48, 33, 16, 8, 'tanh', False),     # widest bank conv
    (6, 43, 2, 32, 31, 15, None, False),       # location conv of the attention: 2 channels, taps folded into columns
    (3, 20, 5, 70, 3, 1, 'relu', False),       # few channels, ragged
])
def test_conv_backward(dev, B, T, Cin, N, KT, pad, act, pool):
    from semi_tts_amd import autograd as AG
    x, w, b = rnd(B, T, Cin, seed=1), rnd(N, Cin, KT, scale=(Cin * KT) ** -0.5, seed=2), rnd(N, seed=3)
    for Tout in sorted({T + 2 * pad - KT + 1, T}):
        if Tout > T + 2 * pad - KT + 1:
            continue
        dy = rnd(B, Tout, N, seed=4)
        xd, wd, bd = (t.to(dev).requires_grad_() for t in (x, w, b))
        y = AG.conv(xd, wd, bd, pad=pad, Tout=Tout, act=act, pool_prev=pool)
        y.backward(dy.to(dev))
        xr, wr, br = (t.double().requires_grad_() for t in (x, w, b))
        yr = conv_cl_ref(xr, wr, br, pad, Tout, act, pool)
        yr.backward(dy.double())
        errs = dict(y=maxdiff(y, yr), dx=relerr(xd.grad, xr.grad), dw=relerr(wd.grad, wr.grad), db=relerr(bd.grad, br.grad))
        report('conv_backward', B=B, T=T, Cin=Cin, N=N, KT=KT, Tout=Tout, **errs)
        assert errs['y'] < 2e-5
        assert max(errs['dx'], errs['dw'], errs['db']) < 2e-5      # fp32 sums of <= B*T*KT*N terms vs float64


def test_linear_backward_with_mask_and_residual(dev):
    from semi_tts_amd import autograd as AG
    M, K, N = 300, 70, 50
    x, w = rnd(M, K, seed=1), rnd(N, K, scale=K ** -0.5, seed=2)
    mask = (torch.rand(M, N) > 0.5).float() * 2
    dy = rnd(M, N, seed=3)
    xd, wd = x.to(dev).requires_grad_(), w.to(dev).requires_grad_()
    y = AG.linear(xd, wd, None, 'relu', mask.to(dev))
    y.backward(dy.to(dev))
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    yr = torch.relu(xr @ wr.t()) * mask.double()
    yr.backward(dy.double())
    assert maxdiff(y, yr) < 1e-5
    assert relerr(xd.grad, xr.grad) < 1e-5 and relerr(wd.grad, wr.grad) < 1e-5
    # residual after a linear map (pre_highway_proj + inputs, src/module.py:607-609)
    res = rnd(2, 150, N, seed=5)
    xd, wd, rd = x.view(2, 150, K).to(dev).requires_grad_(), w.to(dev).requires_grad_(), res.to(dev).requires_grad_()
    y = AG.conv(xd, wd, None, res=rd)
    y.backward(dy.view(2, 150, N).to(dev))
    xr, wr, rr = x.view(2, 150, K).double().requires_grad_(), w.double().requires_grad_(), res.double().requires_grad_()
    (xr @ wr.t() + rr).backward(dy.view(2, 150, N).double())
    assert relerr(xd.grad, xr.grad) < 1e-5 and relerr(wd.grad, wr.grad) < 1e-5 and relerr(rd.grad, rr.grad) < 1e-6


@pytest.mark.parametrize('act', [None, 'relu', 'tanh'])
def test_batchnorm_train_backward(dev, act):
    from semi_tts_amd import autograd as AG
    B, T, N = 3, 41, 70
    x = rnd(B, T, N, seed=1) * 2 + 0.5
    bn = torch.nn.BatchNorm1d(N, momentum=0.99, eps=1e-3)
    with torch.no_grad():
        bn.weight.copy_(rnd(N, seed=2) * 0.3 + 1)
        bn.bias.copy_(rnd(N, seed=3) * 0.3)
    import copy
    bnd = copy.deepcopy(bn).to(dev)
    dy = rnd(B, T, N, seed=4)
    xd = x.to(dev).requires_grad_()
    y = AG.batch_norm_train(xd, bnd, act)
    y.backward(dy.to(dev))
    bn = bn.double()
    xr = x.double().requires_grad_()
    yr = bn(xr.transpose(1, 2)).transpose(1, 2)
    yr = {'relu': torch.relu, 'tanh': torch.tanh, None: lambda v: v}[act](yr)
    yr.backward(dy.double())
    errs = dict(y=maxdiff(y, yr), dx=relerr(xd.grad, xr.grad), dw=relerr(bnd.weight.grad, bn.weight.grad),
                db=relerr(bnd.bias.grad, bn.bias.grad), rm=maxdiff(bnd.running_mean, bn.running_mean),
                rv=maxdiff(bnd.running_var, bn.running_var))
    report('bn_train_backward', act=str(act), **errs)
    assert errs['y'] < 1e-5 and errs['rm'] < 1e-5 and errs['rv'] < 1e-5
    assert max(errs['dx'], errs['dw'], errs['db']) < 2e-5
    assert int(bnd.num_batches_tracked) == 1


@pytest.mark.parametrize('relu_in', [False, True])
def test_batchnorm_bank_against_torch(dev, relu_in):
    """K BatchNorm1d layers of a conv bank as three launches per phase: segments of T and T + 1 frames (the statistics see every row,
    the bank keeps the first T), the ReLU in front of the norm folded into the backward, running statistics of every layer updated --
    against float64 nn.BatchNorm1d + torch.cat.  ref: src/module.py:590-598"""
    import copy
    from semi_tts_amd import autograd as AG
    B, T, N, K = 3, 37, 24, 5
    Ts = [T + 1 if (k + 1) % 2 == 0 else T for k in range(K)]
    pres = [rnd(B, Ts[k], N, seed=10 + k) * 1.5 + 0.2 for k in range(K)]
    bns = [torch.nn.BatchNorm1d(N, momentum=0.99, eps=1e-3) for _ in range(K)]
    with torch.no_grad():
        for k, bn in enumerate(bns):
            bn.weight.copy_(rnd(N, seed=30 + k) * 0.3 + 1)
            bn.bias.copy_(rnd(N, seed=40 + k) * 0.3)
    bnd = [copy.deepcopy(bn).to(dev) for bn in bns]
    dy = rnd(B, T, K * N, seed=5)
    pd = [p.to(dev).requires_grad_() for p in pres]
    xs = [torch.relu(p) for p in pd] if relu_in else pd
    if relu_in:
        # the bank returns gradients at the PRE-activations: hand it the ReLU outputs as leaves and compare with the chain rule below
        xs = [torch.relu(p.detach()).requires_grad_() for p in pd]
    bank = AG.batch_norm_bank(xs, bnd, T, relu_in)
    bank.backward(dy.to(dev))
    pr = [p.double().requires_grad_() for p in pres]
    bref = [bn.double() for bn in bns]
    yr = torch.cat([bref[k]((torch.relu(pr[k]) if relu_in else pr[k]).transpose(1, 2)).transpose(1, 2)[:, :T] for k in range(K)], -1)
    yr.backward(dy.double())
    assert bank.shape == (B, T, K * N) and maxdiff(bank, yr) < 1e-5
    for k in range(K):
        assert relerr(xs[k].grad, pr[k].grad) < 2e-5, k          # (relu_in: the gradient at the conv's pre-activation)
        assert relerr(bnd[k].weight.grad, bref[k].weight.grad) < 2e-5 and relerr(bnd[k].bias.grad, bref[k].bias.grad) < 2e-5
        assert maxdiff(bnd[k].running_mean, bref[k].running_mean) < 1e-5 and maxdiff(bnd[k].running_var, bref[k].running_var) < 1e-5
        assert int(bnd[k].num_batches_tracked) == 1


def test_highway_layer_backward_against_torch(dev):
    """one Highway layer with its two Linear layers as a single N = 2C product (training path) vs float64 torch"""
    from semi_tts_amd.module import Highway
    Cn = 48
    hw = Highway(Cn, Cn)
    with torch.no_grad():
        for i, p_ in enumerate(hw.parameters()):
            p_.copy_(rnd(*p_.shape, seed=60 + i) * 0.3)
    x = rnd(4, 19, Cn, seed=7)
    dy = rnd(4, 19, Cn, seed=8)
    import copy
    hd = copy.deepcopy(hw).to(dev).train()
    xd = x.to(dev).requires_grad_()
    y = hd(xd)
    y.backward(dy.to(dev))
    xr = x.double().requires_grad_()
    hr = copy.deepcopy(hw).double()
    Hh, Tt = torch.relu(hr.H(xr)), torch.sigmoid(hr.T(xr))
    yr = Hh * Tt + xr * (1 - Tt)
    yr.backward(dy.double())
    assert maxdiff(y, yr) < 1e-5 and relerr(xd.grad, xr.grad) < 2e-5
    for (n_, pd_), (_, pr_) in zip(hd.named_parameters(), hr.named_parameters()):
        assert relerr(pd_.grad, pr_.grad) < 2e-5, n_
    # the weights move: the side-by-side layout is refreshed with the other cached layouts
    with torch.no_grad():
        for p_ in hd.parameters():
            p_.add_(0.05)
        for p_ in hr.parameters():
            p_.add_(0.05)
    y2 = hd(x.to(dev))
    xr2 = x.double()
    Hh, Tt = torch.relu(hr.H(xr2)), torch.sigmoid(hr.T(xr2))
    assert maxdiff(y2, Hh * Tt + xr2 * (1 - Tt)) < 1e-5


def test_highway_and_gather_backward(dev):
    from semi_tts_amd import autograd as AG
    H, Tg, x, dy = (rnd(4, 30, 48, seed=s) for s in (1, 2, 3, 4))
    Tg = torch.sigmoid(Tg)
    d = [t.to(dev).requires_grad_() for t in (H, Tg, x)]
    y = AG.highway_combine(*d)
    y.backward(dy.to(dev))
    r = [t.double().requires_grad_() for t in (H, Tg, x)]
    yr = r[0] * r[1] + r[2] * (1 - r[1])
    yr.backward(dy.double())
    assert maxdiff(y, yr) < 1e-6
    for a, b in zip(d, r):
        assert relerr(a.grad, b.grad) < 1e-6
    table, idx = rnd(40, 24, seed=5), torch.randint(0, 40, (6, 17))
    td = table.to(dev).requires_grad_()
    out = AG.gather(td, idx.to(dev))
    g = rnd(6, 17, 24, seed=6)
    out.backward(g.to(dev))
    tr = table.double().requires_grad_()
    F.embedding(idx, tr).backward(g.double())
    assert maxdiff(out, F.embedding(idx, table)) == 0.0
    assert relerr(td.grad, tr.grad) < 1e-6


def test_pool_prev_backward_ties(dev):
    # MaxPool1d(2,1,1)[:T] keeps the FIRST maximum of a window: ties send the gradient to x[t-1]
    from semi_tts_amd import ops
    x = torch.tensor([[[1.0], [1.0], [0.5], [2.0], [2.0], [2.0]]])
    dyp = torch.tensor([[[1.0], [10.0], [100.0], [1000.0], [1e4], [1e5]]])
    dx = ops.pool_prev_bwd(dyp.to(dev), x.to(dev))
    xr = x.double().requires_grad_()
    F.max_pool1d(xr.transpose(1, 2), 2, 1, 1)[:, :, :6].backward(dyp.double().transpose(1, 2))
    assert maxdiff(dx, xr.grad) == 0.0


# ------------------------------------------------------------------------------------ sequence layers
@pytest.mark.parametrize('B,T,I,H', [(5, 13, 24, 32), (33, 7, 16, 48), (4, 9, 12, 24)])
def test_bilstm_backward(dev, B, T, I, H):
    """H % 16 == 0: the packed loop (one launch per step for both directions: dgates . W_hh with the pointwise backward of the previous step
    in its epilogue); otherwise the two-launch form on natural operands"""
    from semi_tts_amd import autograd as AG
    lstm = torch.nn.LSTM(I, H, batch_first=True, bidirectional=True).double()
    x = rnd(B, T, I, seed=1)
    dy = rnd(B, T, 2 * H, seed=2)
    p = {k: v.detach().float().to(dev).requires_grad_() for k, v in lstm.named_parameters()}
    xd = x.to(dev).requires_grad_()
    xp_f = AG.conv(xd, p['weight_ih_l0'], p['bias_ih_l0'])
    xp_b = AG.conv(xd, p['weight_ih_l0_reverse'], p['bias_ih_l0_reverse'])
    y = AG.bilstm(xp_f, xp_b, p['weight_hh_l0'], p['bias_hh_l0'], p['weight_hh_l0_reverse'], p['bias_hh_l0_reverse'])
    y.backward(dy.to(dev))
    xr = x.double().requires_grad_()
    yr, _ = lstm(xr)
    yr.backward(dy.double())
    errs = {'y': maxdiff(y, yr), 'dx': relerr(xd.grad, xr.grad)}
    for k, v in lstm.named_parameters():
        errs[k] = relerr(p[k].grad, v.grad)
    report('bilstm_backward', **errs)
    assert errs.pop('y') < 1e-5
    assert max(errs.values()) < 2e-5, errs


@pytest.mark.parametrize('B,T,H', [(3, 17, 20), (4, 33, 80)])
def test_bigru_backward(dev, B, T, H):
    from semi_tts_amd import autograd as AG
    gru = torch.nn.GRU(H, H, batch_first=True, bidirectional=True).double()
    x = rnd(B, T, H, seed=1)
    dy = rnd(B, T, 2 * H, seed=2)
    p = {k: v.detach().float().to(dev).requires_grad_() for k, v in gru.named_parameters()}
    xd = x.to(dev).requires_grad_()
    gi_f = AG.conv(xd, p['weight_ih_l0'], p['bias_ih_l0'])
    gi_b = AG.conv(xd, p['weight_ih_l0_reverse'], p['bias_ih_l0_reverse'])
    y = AG.bigru(gi_f, gi_b, p['weight_hh_l0'], p['bias_hh_l0'], p['weight_hh_l0_reverse'], p['bias_hh_l0_reverse'])
    y.backward(dy.to(dev))
    xr = x.double().requires_grad_()
    yr, _ = gru(xr)
    yr.backward(dy.double())
    errs = {'y': maxdiff(y, yr), 'dx': relerr(xd.grad, xr.grad)}
    for k, v in gru.named_parameters():
        errs[k] = relerr(p[k].grad, v.grad)
    report('bigru_backward', B=B, T=T, H=H, **errs)
    assert errs.pop('y') < 1e-5
    assert max(errs.values()) < 2e-5, errs


# ------------------------------------------------------------------------------------ module level vs the oracle
def oracle_grads(fn, weights, inputs, douts):
    """run an oracle function in float64 on leaf copies of weights/inputs; returns (outputs, weight grads, input grads)"""
    W = {k: (v.double().requires_grad_() if v.is_floating_point() else v) for k, v in weights.items()}
    ins = [t.double().requires_grad_() for t in inputs]
    torch.set_default_dtype(torch.float64)       # the oracle creates its zero states in the default dtype
    try:
        outs = fn(W, *ins)
    finally:
        torch.set_default_dtype(torch.float32)
    outs = outs if isinstance(outs, (tuple, list)) else (outs,)
    pairs = [(o, d) for o, d in zip(outs, douts) if d is not None]
    torch.autograd.backward([o for o, _ in pairs], [d.double() for _, d in pairs])
    return outs, {k: v.grad for k, v in W.items() if v.is_floating_point() and v.grad is not None}, [t.grad for t in ins]


def check_param_grads(module, prefix, wg, tol, name):
    worst = 0.0
    n = 0
    for k, p in module.named_parameters():
        ref = wg.get(prefix + k)
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, 'unexpected gradient for ' + k
            continue
        assert p.grad is not None, 'missing gradient for ' + k
        if float(ref.abs().max()) < 1e-9:
            # analytically zero (a bias in front of a BatchNorm): float64 leaves ~1e-16, fp32 ~1e-6 of round-off
            assert float(p.grad.abs().max()) < 1e-4, (k, float(p.grad.abs().max()))
            continue
        e = relerr(p.grad, ref)
        report(name, param=k, err=e)
        worst = max(worst, e)
        n += 1
        assert e < tol, (k, e)
    assert n > 0
    return worst


def test_encoder_backward_vs_oracle(dev):
    from conftest import load_golden
    from helpers import tiny_tacotron
    from oracle import tts_oracle as O
    W, A, meta = load_golden('tts_tiny_train_tf')
    m = tiny_tacotron(meta, W, dev).train()
    x = A['txt_embed'].float()
    xd = x.to(dev).requires_grad_()
    out = m.encoder(xd)
    dy = rnd(*out.shape, seed=5)
    out.backward(dy.to(dev))
    outs, wg, ig = oracle_grads(lambda Wd, xx: O.encoder_forward(Wd, xx, 'encoder.', training=True), W, [x], [dy])
    assert maxdiff(out, outs[0]) < 1e-5
    assert relerr(xd.grad, ig[0]) < 5e-5
    check_param_grads(m.encoder, 'encoder.', wg, 5e-5, 'encoder_backward')


def test_postnet_backward_vs_oracle(dev):
    from conftest import load_golden
    from helpers import tiny_tacotron
    from oracle import tts_oracle as O
    W, A, meta = load_golden('tts_tiny_train_tf')
    m = tiny_tacotron(meta, W, dev).train()
    n_mels = meta['cfg']['n_mels']
    mel = torch.rand(3, 21, n_mels, generator=torch.Generator().manual_seed(7))
    md = mel.to(dev).requires_grad_()
    m.zero_grad()
    lin = m.postnet(md)
    dy = rnd(*lin.shape, seed=9)
    lin.backward(dy.to(dev))
    outs, wg, ig = oracle_grads(lambda Wd, xx: O.postnet_forward(Wd, xx, training=True), W, [mel], [dy])
    assert maxdiff(lin, outs[0]) < 2e-5
    assert relerr(md.grad, ig[0]) < 1e-4
    check_param_grads(m.postnet, 'postnet.', wg, 1e-4, 'postnet_backward')      # BN-heavy chain: fp32 vs float64


# ------------------------------------------------------------------------------------ decoder BPTT / whole model
def _double_masks(masks):
    return [m.double() for m in masks]


@pytest.mark.parametrize('name', ['tts_tiny_train_tf', 'tts_tiny_sched', 'tts_tiny_partial', 'tts_tiny_pretrain', 'tts_tiny_dropin',
                                  'tts_tiny_noloc', 'tts_tiny_nosum', 'tts_tiny_encdrop', 'tts_tiny_preln_train', 'tts_tiny_prebn_train',
                                  # own-output feedback through a normalised prenet (LayerNorm; BatchNorm1d over the fed-back rows of a step)
                                  'tts_tiny_preln_sched', 'tts_tiny_prebn_sched', 'tts_tiny_prebn_partial'])
def test_tacotron2_backward_against_oracle_tiny_golden(dev, name):
    """Whole Tacotron2 in training mode with the reference's recorded dropout masks and coin flips -- teacher forcing,
    scheduled sampling (own output fed back on some steps) and a partial-teacher batch (unpaired rows always feed their
    own output back): every parameter gradient and the input gradients vs float64 autograd through the oracle."""
    from conftest import load_golden
    from helpers import coin_source, masks_to, split_masks, tiny_tacotron
    from oracle import tts_oracle as O
    from semi_tts_amd.module import plan_decode
    W, A, meta = load_golden(name)
    hp = meta['hp']
    m = tiny_tacotron(meta, W, dev).train()
    txt, spk, teacher = A['txt_embed'], A['spkr_embed'], A['teacher']
    B = txt.shape[0]
    steps, src = plan_decode(False, teacher.shape[1], teacher.shape[0], B, hp['n_frames_per_step'], meta['tf_rate'],
                             hp['drop_dec_in'], meta['unpair_max_frame'], coin_source(A['coins']))
    masks = masks_to(split_masks(A.get('mask', []), hp, True, meta['tf_rate'], B, teacher.shape[0], steps, src,
                                 hp['prenet_dim']), dev)
    import numpy as np
    saved = np.random.rand
    np.random.rand = coin_source(A['coins'])
    try:
        txt_d, spk_d = txt.to(dev).requires_grad_(), spk.to(dev).requires_grad_()
        mel, lin, align, stop = m(txt_d, None, teacher.to(dev), spk_d, tf_rate=meta['tf_rate'],
                                  unpair_max_frame=meta['unpair_max_frame'], _masks=masks)
    finally:
        np.random.rand = saved
    douts = [rnd(*mel.shape, seed=1), rnd(*lin.shape, seed=2), rnd(*align.shape, seed=3), rnd(*stop.shape, seed=4)]
    if hp.get('pretrain'):
        douts[2] = None                 # the alignments are constant zeros without attention (src/module.py:238-240)
    torch.autograd.backward([o for o, d in zip((mel, lin, align, stop), douts) if d is not None],
                            [d.to(dev) for d in douts if d is not None])

    def fn(Wd, t, s):
        drop = O.DropoutSource('list', _double_masks(A.get('mask', [])))
        return O.tacotron2_forward(Wd, t, teacher.double(), s, hp, meta['tf_rate'], meta['unpair_max_frame'], True, drop,
                                   coin_source(A['coins']))
    outs, wg, ig = oracle_grads(fn, W, [txt, spk], douts)
    assert maxdiff(mel, outs[0]) < 1e-4 and maxdiff(lin, outs[1]) < 2e-4
    if ig[0] is None:                   # pre-training: nothing downstream reads the encoder output
        assert txt_d.grad is None or float(txt_d.grad.abs().max()) < 1e-6
        ig[0] = torch.zeros_like(txt).double()
        txt_d.grad = torch.zeros_like(txt_d)
    errs = dict(dtxt=float((txt_d.grad.cpu().double() - ig[0]).abs().max()) if float(ig[0].abs().max()) == 0.0 else relerr(txt_d.grad, ig[0]),
                dspk=relerr(spk_d.grad, ig[1]))
    report('tacotron2_backward_tiny', name=name, **errs)
    assert errs['dtxt'] < 2e-4 and errs['dspk'] < 2e-4
    check_param_grads(m, '', wg, 2e-4, 'tacotron2_backward_tiny')     # fp32 BPTT over 4 steps + BN chains vs float64


@pytest.mark.parametrize('B,L,steps,tiled', [(20, 11, 7, False), (3, 5, 3, False), (20, 11, 7, True), (33, 6, 4, True)])
def test_decoder_backward_against_oracle(dev, B, L, steps, tiled):
    """Decoder only, dropout masks drawn by the oracle and replayed through the HIP path; B=20 exercises the padded rows of the step
    tapes (Bp=32).  tiled=False: dimensions that are not multiples of the tile sizes (six launches per backward step);
    tiled=True: dimensions the fused form takes -- the cells' pointwise backward in the epilogues of the loop's products."""
    from helpers import masks_to, split_masks
    from oracle import tts_oracle as O
    from semi_tts_amd.module import Decoder
    hp = dict(n_frames_per_step=2, prenet_dim=24, prenet_dropout=0.5, query_rnn_dim=40, dec_rnn_dim=36, query_dropout=0.1,
              dec_dropout=0.1, attn_dim=32, n_location_filters=8, location_kernel_size=7, loc_aware=True,
              use_summed_weights=True, drop_dec_in=0.0)
    n_mels, E, S = 10, 28, 12
    if tiled:
        hp.update(prenet_dim=32, query_rnn_dim=48, dec_rnn_dim=64)
        E = 32
    torch.manual_seed(5)
    dec = Decoder(n_mels, enc_embed_dim=E, spkr_embed_dim=S, **hp).to(dev).train()
    W = {'decoder.' + k: v.detach().cpu() for k, v in dec.state_dict().items()}
    memory, spk = rnd(B, L, E, seed=1), rnd(B, S, seed=2)
    teacher = torch.rand(B, steps * 2, n_mels, generator=torch.Generator().manual_seed(3))
    hpo = dict(hp, n_mels=n_mels)
    drop = O.DropoutSource('rng', generator=torch.Generator().manual_seed(11))
    torch.set_default_dtype(torch.float64)
    try:
        Wd = {k: v.double().requires_grad_() for k, v in W.items()}
        mem_r, spk_r = memory.double().requires_grad_(), spk.double().requires_grad_()

        class Drop64(O.DropoutSource):
            def __call__(self, x, p, training):
                if (not training) or p == 0.0:
                    return x
                keep = torch.full(x.shape, 1.0 - p, dtype=torch.float32)
                mk = torch.bernoulli(keep, generator=self.gen) / (1.0 - p)
                self.used.append(mk)
                return x * mk.double()
        drop = Drop64('rng', generator=torch.Generator().manual_seed(11))
        outs = O.decoder_forward(Wd, mem_r, teacher.double(), spk_r, hpo, 1.0, None, True, drop, lambda: 0.0)
    finally:
        torch.set_default_dtype(torch.float32)
    douts = [rnd(*outs[0].shape, seed=1), rnd(*outs[1].shape, seed=2), rnd(*outs[2].shape, seed=3)]
    torch.autograd.backward(list(outs), [d.double() for d in douts])
    src = list(range(steps))
    masks = masks_to(split_masks(drop.used, hpo, True, 1.0, B, B, steps, src, hp['prenet_dim']), dev)
    mem_d, spk_d = memory.to(dev).requires_grad_(), spk.to(dev).requires_grad_()
    mel, align, stop = dec(mem_d, None, teacher.to(dev), spk_d, tf_rate=1.0, _masks=masks)
    torch.autograd.backward([mel, align, stop], [d.to(dev) for d in douts])
    errs = dict(mel=maxdiff(mel, outs[0]), align=maxdiff(align, outs[1]), dmem=relerr(mem_d.grad, mem_r.grad),
                dspk=relerr(spk_d.grad, spk_r.grad))
    report('decoder_backward', B=B, L=L, steps=steps, **errs)
    assert errs['mel'] < 1e-4 and errs['align'] < 1e-5
    assert errs['dmem'] < 1e-4 and errs['dspk'] < 1e-4
    wg = {k: v.grad for k, v in Wd.items() if v.grad is not None}
    check_param_grads(dec, 'decoder.', wg, 1e-4, 'decoder_backward')


# ------------------------------------------------------------------------------------ H1: the training step
def test_freq_loss_against_reference_golden(dev):
    from conftest import load_golden
    from semi_tts_amd import autograd as AG
    _, A, _ = load_golden('freq_loss')
    for pred, lab, ref, kind in ((A['pm'], A['lm'], A['mel_mse'], 'mse'), (A['pl'], A['ll'], A['lin_mse'], 'mse'),
                                 (A['pl'], A['ll'], A['lin_l1'], 'l1')):
        pd = pred.to(dev).requires_grad_()
        loss = AG.freq_loss(pd, lab.to(dev), 22050, 80, kind, True, True)
        (loss * 3.0).backward()
        from oracle import tts_oracle as O
        pr = pred.double().requires_grad_()
        (O.freq_loss(pr, lab.double(), 22050, 80, kind, True, True) * 3.0).backward()
        assert abs(float(loss.detach()) - float(ref)) < 1e-6 * max(1.0, abs(float(ref)))
        assert relerr(pd.grad, pr.grad) < 1e-5


def test_training_step_against_reference_golden(dev):
    """Two optimisation steps of the paired TTS branch (VQVAE.text_to_speech -> freq_loss -> backward -> clip ->
    Adam with the 'decay' schedule) against what the real reference produced for the same weights, batch and
    dropout masks: losses, grad norm, every parameter gradient of step 1, selected parameters after step 2."""
    import json
    from argparse import Namespace
    from conftest import load_golden
    from helpers import masks_to, split_masks, tiny_vqvae
    from semi_tts_amd.solver import TtsTrainer
    W, A, meta = load_golden('train_step_tiny')
    hp = meta['hp']
    config = dict(data=dict(audio=meta['audio'], corpus=dict(batch_size=4)), hparas=meta['hparas'], model=meta['model'])
    tr = TtsTrainer(config, Namespace(vocab_size=meta['vocab_size'], n_spkr=meta['n_spkr'], verbose=False, max_step=2), 'train')
    tr.model = tiny_vqvae(meta, W, dev).train()
    from semi_tts_amd.optim import Optimizer
    h = meta['hparas']
    tr.optimizer = Optimizer(tr.model.parameters(), h['optimizer'], h['lr'], h['lr_scheduler'], tf_start=h['tf_start'],
                             tf_end=h['tf_end'], tf_step=h['tf_step'])
    text, sid, mel, linear = (A[k].to(dev) for k in ('text', 'sid', 'mel', 'linear'))
    B, steps = text.shape[0], mel.shape[1] // hp['n_frames_per_step']
    pos = 0
    for step, (n, ref) in enumerate(zip(meta['n_masks'], meta['stats'])):
        rec = A['mask'][pos:pos + n]
        pos += n
        masks = masks_to(split_masks(rec, hp, True, 1.0, B, B, steps, list(range(steps)), hp['prenet_dim']), dev)
        if step == 0:
            # gradients of the first backward, before clipping
            grads = {}
            hooks = []
            st = None
            orig = tr.clip_grad_norm_

            def spy(params, max_norm):
                params = list(params)
                for (kname, p) in tr.model.named_parameters():
                    if p.grad is not None:
                        grads[kname] = p.grad.detach().clone()
                return orig(params, max_norm)
            tr.clip_grad_norm_ = spy
            try:
                st = tr.train_step(text, sid, mel, linear, _masks=masks)
            finally:
                del tr.clip_grad_norm_
            keys = json.loads(bytes(A['grad_keys']).decode())
            worst = 0.0
            for k, gref in zip(keys, A['grad']):
                assert k in grads, 'missing gradient for ' + k
                if float(gref.abs().max()) < 1e-9:
                    assert float(grads[k].abs().max()) < 1e-5
                    continue
                e = relerr(grads[k], gref)
                worst = max(worst, e)
                assert e < 5e-4, (k, e)          # fp32 HIP vs fp32 reference (different summation orders on both sides)
            report('train_step_grads', worst=worst, n=len(keys))
        else:
            st = tr.train_step(text, sid, mel, linear, _masks=masks)
        report('train_step', step=step, loss=st['loss'], ref_loss=ref['loss'], gn=st['grad_norm'], ref_gn=ref['grad_norm'])
        assert abs(st['loss'] - ref['loss']) < 2e-5 and abs(st['mel_loss'] - ref['mel_loss']) < 2e-5
        assert abs(st['grad_norm'] - ref['grad_norm']) < 1e-4 * ref['grad_norm']
        assert abs(st['lr'] - ref['lr']) < 1e-12 and st['tf_rate'] == ref['tf_rate']
    sd = tr.model.state_dict()
    for k, v in zip(json.loads(bytes(A['post_keys']).decode()), A['post']):
        # Adam's first steps move every weight by ~lr (1e-6, 2e-6) whatever the gradient scale: the bound checks
        # the update direction element-wise wherever the gradient is not round-off
        assert maxdiff(sd[k], v) < 2e-5, k
        if 'running_' not in k and 'num_batches' not in k:
            upd_ref = (v.double() - W[k].double())
            upd = (sd[k].detach().cpu().double() - W[k].double())
            assert float((upd - upd_ref).abs().mean()) < 0.05 * float(upd_ref.abs().mean()), k


def test_multi_tensor_clip_and_adam_against_torch(dev):
    """clip_grad_norm_ + Adam on the multi-tensor kernels vs torch's own implementations on CPU, over tensors that span
    several launches (more than 24 tensors, one larger than 320 chunks of 64K) and 3 steps with changing lr."""
    from semi_tts_amd.optim import FusedAdam, clip_grad_norm_
    g = torch.Generator().manual_seed(0)
    shapes = [(7,), (33, 5), (1, 1), (128, 257)] * 7 + [(21 * 1024 * 1024 + 11,)]
    ref = [torch.randn(*s, generator=g).requires_grad_() for s in shapes]
    hip = [p.detach().clone().to(dev).requires_grad_() for p in ref]
    o_ref, o_hip = torch.optim.Adam(ref, lr=1e-3), FusedAdam(hip, lr=1e-3)
    for step in range(3):
        for grp in o_ref.param_groups + o_hip.param_groups:
            grp['lr'] = 1e-3 * (step + 1)
        for pr, ph in zip(ref, hip):
            gr = torch.randn(pr.shape, generator=g) * (10.0 if step == 1 else 0.01)     # step 1 clips, the others do not
            pr.grad, ph.grad = gr.clone(), gr.clone().to(dev)
        # float64 restatement of torch.nn.utils.clip_grad_norm_ (torch's own fp32 norm of a 22 M-element tensor is
        # itself 1e-3 off the float64 value, so it cannot be the yard-stick)
        n_ref = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in ref))
        coef = min(1.0, 5.0 / (float(n_ref) + 1e-6))
        for p in ref:
            p.grad.mul_(coef)
        n_hip = clip_grad_norm_(hip, 5.0)
        assert abs(float(n_hip) - float(n_ref)) < 2e-6 * float(n_ref)
        assert max(relerr(ph.grad, pr.grad) for pr, ph in zip(ref, hip)) < 1e-5
        o_ref.step(); o_hip.step()
        for pr, ph in zip(ref, hip):
            assert maxdiff(ph, pr) < 2e-6      # fp32 update arithmetic in a different association
    sd = o_hip.state_dict()
    assert set(sd['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'} and float(sd['state'][0]['step']) == 3.0
    o2 = FusedAdam(hip, lr=1e-3)
    o2.load_state_dict(sd)
    assert o2.state[hip[0]]['_step'] == 3


def test_optimiser_passes_table_form_equals_kernel_argument_form(dev):
    """The gradient norm / clip / Adam launches read their tensor list from a cached device-side table (one launch per pass); during a
    stream capture no table can be made and the list travels in the kernel arguments (several launches): same blocks in the same order,
    so the two forms must agree bit for bit -- here the second form is run by capturing the calls into a hipGraph and replaying it."""
    from semi_tts_amd import ops, _lib
    from semi_tts_amd.optim import FusedAdam, clip_grad_norm_
    g = torch.Generator().manual_seed(3)
    shapes = [(7,), (33, 5), (1, 1), (128, 257)] * 9 + [(3 * 1024 * 1024 + 5,)]
    base = [torch.randn(*s, generator=g) for s in shapes]
    grads = [torch.randn(*s, generator=g) * 3.0 for s in shapes]

    def fresh():
        ps = [b.clone().to(dev).requires_grad_() for b in base]
        for p, gr in zip(ps, grads):
            p.grad = gr.clone().to(dev)
        return ps, FusedAdam(ps, lr=1e-3)

    lib = _lib.load()
    pa, oa = fresh()
    misses = lib.st_mt_table_misses()
    na = clip_grad_norm_(pa, 5.0)
    oa.step()
    assert lib.st_mt_table_misses() > misses               # (the table form ran: new addresses, new tables)
    torch.cuda.synchronize()
    pb, ob = fresh()
    gr = ops.Graph()
    with gr.memory():
        ob.step()                                          # (moments and workspaces exist before the capture; undone below)
        clip_grad_norm_(pb, 5.0)
    torch.cuda.synchronize()
    with torch.no_grad():
        for p, b, g0 in zip(pb, base, grads):
            p.copy_(b.to(dev)); p.grad.copy_(g0.to(dev))
        for st in ob.state.values():
            for k in ('exp_avg', 'exp_avg_sq'):
                st[k].zero_()
            st['_step'] = 0
    held = {}
    misses = lib.st_mt_table_misses()
    with gr.capture():
        with gr.memory():
            held['n'] = clip_grad_norm_(pb, 5.0)
            ob.step()
    assert lib.st_mt_table_misses() == misses              # (no table during the capture)
    gr.launch()
    torch.cuda.synchronize()
    assert torch.equal(held['n'], na)
    for a, b in zip(pa, pb):
        assert torch.equal(a, b) and torch.equal(a.grad, b.grad)


def test_clip_with_pre_scale_and_multi_tensor_copy(dev):
    """the data-parallel forms of the optimiser helpers: clip_grad_norm_(pre_scale = 1 / world) on gradients that are still the SUM over
    the ranks = the plain clip on the averaged gradients (norm and result); ops.mt_copy gathers stray gradients into bucket slots
    (aligned and unaligned, one element up to several chunks, more tensors than one launch takes)"""
    from semi_tts_amd import ops
    from semi_tts_amd.optim import clip_grad_norm_
    g = torch.Generator().manual_seed(5)
    sizes = [1, 3, 80, 4097, 40000, 1025 * 160] + [17 + i for i in range(120)]
    srcs = [torch.randn(n, generator=g).to(dev) for n in sizes]
    flat = torch.full((sum(sizes) + 8,), float('nan'), device=dev)
    dsts, off = [], 1                                            # (slots packed back to back from an odd offset: every alignment occurs)
    for n in sizes:
        dsts.append(flat[off:off + n])
        off += n
    ops.mt_copy(dsts, srcs)
    for d, s_ in zip(dsts, srcs):
        assert torch.equal(d, s_)
    assert bool(torch.isnan(flat[0])) and bool(torch.isnan(flat[off:]).all())
    for world, max_norm in ((4, 5.0), (4, 0.05), (3, 0.05)):
        class P:                                                 # (clip_grad_norm_ only reads .grad)
            pass
        avg, summed = [], []
        for s_ in srcs[:40]:
            a, b = P(), P()
            a.grad, b.grad = (s_ / world).clone(), s_.clone()
            avg.append(a)
            summed.append(b)
        n_ref = clip_grad_norm_(avg, max_norm)
        n_got = clip_grad_norm_(summed, max_norm, pre_scale=1.0 / world)
        assert abs(float(n_got) - float(n_ref)) <= 2e-6 * float(n_ref)
        for a, b in zip(avg, summed):
            assert maxdiff(a.grad, b.grad) <= 2e-6 * max(1.0, float(a.grad.abs().max()))


def test_checkpoint_round_trip_resumes_training(dev, tmp_path):
    """{"model", "optimizer", "global_step"} (the reference's checkpoint layout, src/solver.py:203-216): train 2 steps,
    save, reload into a fresh trainer, and the third step must be bit-identical to continuing in place."""
    from argparse import Namespace
    from conftest import load_golden
    from helpers import tiny_vqvae
    from semi_tts_amd.optim import Optimizer
    from semi_tts_amd.solver import TtsTrainer
    W, A, meta = load_golden('train_step_tiny')
    config = dict(data=dict(audio=meta['audio'], corpus=dict(batch_size=4)), hparas=dict(meta['hparas'], lr_scheduler='fixed'),
                  model=meta['model'])
    h = config['hparas']
    batch = [A[k].to(dev) for k in ('text', 'sid', 'mel', 'linear')]

    def make(load=None):
        tr = TtsTrainer(config, Namespace(vocab_size=meta['vocab_size'], n_spkr=meta['n_spkr'], verbose=False, max_step=3,
                                          ckpdir=str(tmp_path), name='rt', load=load), 'train')
        tr.model = tiny_vqvae(meta, W, dev).train()
        for mod in tr.model.modules():                 # no dropout: the two runs must see identical arithmetic
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        tr.model.tts.decoder.prenet_dropout = 0.0
        tr.model.tts.decoder.prenet.apply_dropout = 0.0
        tr.optimizer = Optimizer(tr.model.parameters(), h['optimizer'], h['lr'], h['lr_scheduler'])
        if load:
            tr.load_ckpt()
        return tr
    a = make()
    for _ in range(2):
        a.train_step(*batch)
    path = a.save_checkpoint('two.pth', 0.0)
    ck = torch.load(path, map_location='cpu')
    assert set(ck) == {'model', 'optimizer', 'global_step'} and ck['global_step'] == 2
    assert set(ck['optimizer']) == {'state', 'param_groups'}
    st_a = a.train_step(*batch)
    b = make(load=path)
    assert b.step == 2
    st_b = b.train_step(*batch)
    assert st_a['loss'] == st_b['loss'] and st_a['grad_norm'] == st_b['grad_norm']      # the whole step is deterministic
    sa, sb = a.model.state_dict(), b.model.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    assert a.optimizer.opt.state[next(iter(a.model.tts.parameters()))]['_step'] == 3


# ------------------------------------------------------------------------------------ codebook lookup / speech encoder backward
@pytest.mark.parametrize('B,S,D,V,attr,n_real', [(3, 7, 64, 43, True, 0), (4, 9, 64, 512, False, 0), (5, 6, 64, 43, False, 2),
                                                  (2, 5, 24, 10, False, 0)])
def test_vq_l2_backward_vs_oracle(dev, B, S, D, V, attr, n_real):
    """L2Embedding.forward in training mode: straight-through gradient to the latents, p_code gradient to latents and table
    (only the first `first_n_real_mel` utterances reach the table when set), scatter-add of the picked rows -- against
    float64 autograd through the oracle (src/embed.py:105-147)."""
    from oracle import vq_oracle as VQ
    from semi_tts_amd.embed import L2Embedding
    torch.manual_seed(B * 100 + V)
    cb = L2Embedding(V, False, 'normal', D, 0, 0, 1, 0, True)
    W = {'learnable_table': rnd(V, D - (16 if attr else 0), scale=0.5, seed=1), 'temp': torch.tensor([0.7])}
    if attr:
        cb.use_phn_attr = True
        cb.phn_attr = torch.nn.Embedding.from_pretrained((torch.rand(V, 31) > 0.5).float(), freeze=True, padding_idx=0)
        cb.proj_attr = torch.nn.Linear(31, 16)
        W.update({'phn_attr.weight': cb.phn_attr.weight.detach().clone(), 'proj_attr.weight': cb.proj_attr.weight.detach().clone(),
                  'proj_attr.bias': cb.proj_attr.bias.detach().clone()})
    cb.learnable_table = torch.nn.Parameter(W['learnable_table'].clone())
    cb.temp.copy_(W['temp'])
    cb = cb.to(dev).train()
    x = rnd(B, S, D, scale=0.6, seed=2)
    dp, dl = rnd(B, S, V, seed=3), rnd(B, S, D, seed=4)
    xd = x.to(dev).requires_grad_()
    p, lat, _, _ = cb(xd, n_real)
    torch.autograd.backward([p, lat], [dp.to(dev), dl.to(dev)])

    def fn(Wd, xx):
        pp, idx, nl, _ = VQ.l2_forward(Wd, xx, n_real)
        return pp, nl
    outs, wg, ig = oracle_grads(fn, W, [x], [dp, dl])
    assert torch.equal(cb.last_idx.cpu(), outs[0].argmax(-1))
    errs = dict(p=maxdiff(p, outs[0]), lat=maxdiff(lat, outs[1]), dx=relerr(xd.grad, ig[0]),
                dtable=relerr(cb.learnable_table.grad, wg['learnable_table']))
    if attr:
        errs['dproj_w'] = relerr(cb.proj_attr.weight.grad, wg['proj_attr.weight'])
        errs['dproj_b'] = relerr(cb.proj_attr.bias.grad, wg['proj_attr.bias'])
    report('vq_l2_backward', B=B, V=V, **errs)
    assert errs['p'] < 5e-5 and errs['lat'] < 1e-5
    assert all(v < 2e-5 for k, v in errs.items() if k.startswith('d')), errs


@pytest.mark.parametrize('variant', ['st_onehot', 'learn_temp', 'skip', 'st_onehot_real2'])
def test_vq_l2_variants_backward_vs_oracle(dev, variant):
    """The L2Embedding options no shipped config turns on: the ST-onehot code (stop_grad=False, src/embed.py:137-138: the latent
    gradient also reaches p_code), a learnable temperature (temp < 0 in the constructor, :33-34), the skip connection
    (skip_prob, :140-142) -- forward values and every gradient against float64 autograd through the oracle."""
    import numpy as np
    from oracle import vq_oracle as VQ
    from semi_tts_amd.embed import L2Embedding
    B, S, D, V = 4, 6, 32, 21
    n_real = 2 if variant.endswith('real2') else 0
    stop = not variant.startswith('st_onehot')
    cb = L2Embedding(V, False, 'normal', D, 0, 0, -1 if variant == 'learn_temp' else 1, 1.0 if variant == 'skip' else 0, stop)
    W = {'learnable_table': rnd(V, D, scale=0.5, seed=1), 'temp': torch.tensor([0.7])}
    cb.learnable_table = torch.nn.Parameter(W['learnable_table'].clone())
    with torch.no_grad():
        cb.temp.copy_(W['temp'])
    cb = cb.to(dev).train()
    x = rnd(B, S, D, scale=0.6, seed=2)
    dp, dl = rnd(B, S, V, seed=3), rnd(B, S, D, seed=4)
    xd = x.to(dev).requires_grad_()
    p, lat, _, _ = cb(xd, n_real)
    torch.autograd.backward([p, lat], [dp.to(dev), dl.to(dev)])

    def fn(Wd, xx):
        pp, idx, nl, _ = VQ.l2_forward(Wd, xx, n_real, stop_grad=stop, skip=variant == 'skip')
        return pp, nl
    outs, wg, ig = oracle_grads(fn, W, [x], [dp, dl])
    assert torch.equal(cb.last_idx.cpu(), outs[0].argmax(-1))
    errs = dict(p=maxdiff(p, outs[0]), lat=maxdiff(lat, outs[1]), dx=relerr(xd.grad, ig[0]),
                dtable=relerr(cb.learnable_table.grad, wg['learnable_table']))
    if variant == 'learn_temp':
        assert isinstance(cb.temp, torch.nn.Parameter)
        errs['dtemp'] = abs(float(cb.temp.grad) - float(wg['temp'])) / max(1.0, abs(float(wg['temp'])))
    report('vq_l2_variants', variant=variant, **errs)
    assert errs['p'] < 5e-5 and errs['lat'] < 1e-5
    assert all(v < 3e-5 for k, v in errs.items() if k.startswith('d')), errs


@pytest.mark.parametrize('stop_grad', [True, False])
def test_seperate_embedding_backward_vs_oracle(dev, stop_grad):
    """SeperateEmbedding.forward in training mode (the codebook of config/supervised.yaml): softmax(Linear) posterior +
    embedding of the argmax (no straight-through path); stop_grad=False = the ST-onehot code whose gradient also reaches the
    posterior.  ref: src/embed.py:187-205"""
    from oracle import vq_oracle as VQ
    from semi_tts_amd.embed import SeperateEmbedding
    torch.manual_seed(3)
    V, D = 43, 64
    cb = SeperateEmbedding(V, False, 'normal', D, 0, 0, 1, 0, stop_grad).to(dev).train()
    W = {k: v.detach().cpu().clone() for k, v in cb.state_dict().items()}
    x = rnd(3, 8, D, seed=5)
    dp, dl = rnd(3, 8, V, seed=6), rnd(3, 8, D, seed=7)
    xd = x.to(dev).requires_grad_()
    p, lat, _, _ = cb(xd)
    torch.autograd.backward([p, lat], [dp.to(dev), dl.to(dev)])
    outs, wg, ig = oracle_grads(lambda Wd, xx: (lambda r: (r[0], r[2]))(VQ.seperate_forward(Wd, xx, stop_grad)), W, [x], [dp, dl])
    assert maxdiff(p, outs[0]) < 1e-5 and maxdiff(lat, outs[1]) < 1e-6
    assert relerr(xd.grad, ig[0]) < 2e-5
    assert relerr(cb.asr_final_layer.weight.grad, wg['asr_final_layer.weight']) < 2e-5
    assert relerr(cb.asr_final_layer.bias.grad, wg['asr_final_layer.bias']) < 2e-5
    assert relerr(cb.embedding.weight.grad, wg['embedding.weight']) < 2e-5


@pytest.mark.parametrize('gname', ['asr_tiny_train', 'asr_tiny_ln_train'])
def test_ctc_encoder_backward_vs_oracle(dev, gname):
    """The CTC speech encoder in training mode (stride-2 conv layer, batch-statistics BatchNorm + tanh + residual, 2-layer
    BiLSTM, projection) against float64 autograd through the oracle.  ref: src/asr.py:46-64, src/module.py:627-648"""
    from conftest import load_golden
    from oracle import asr_oracle as AO
    from semi_tts_amd.asr import CTC
    W, A, meta = load_golden(gname)
    cfg = meta['cfg']
    m = CTC(meta['in_dim'], meta['out_dim'], **cfg)
    m.load_state_dict(W)
    m = m.to(dev).train()
    x = A['x']
    xd = x.to(dev).requires_grad_()
    y = m(xd)
    assert maxdiff(y, A['y']) < 2e-5                     # the differentiable path gives the reference's training-mode output
    dy = rnd(*y.shape, seed=9)
    y.backward(dy.to(dev))
    outs, wg, ig = oracle_grads(lambda Wd, xx: AO.ctc_forward(Wd, xx, cfg, training=True), W, [x], [dy])
    e = relerr(xd.grad, ig[0])
    report('ctc_encoder_backward', dx=e)
    assert e < 2e-4
    check_param_grads(m, '', wg, 2e-4, 'ctc_encoder_backward')


def test_ctc_encoder_unidirectional_backward_against_reference_golden(dev):
    """`rnn_bid: False` (src/asr.py:35-37): the 2-layer unidirectional LSTM + LayerNorm speech encoder in training mode; output,
    input gradient and EVERY parameter gradient against what the real reference's autograd produced for the recorded dy."""
    from conftest import load_golden
    from semi_tts_amd.asr import CTC
    W, A, meta = load_golden('asr_tiny_uni_train')
    m = CTC(meta['in_dim'], meta['out_dim'], **meta['cfg'])
    m.load_state_dict(W)
    m = m.to(dev).train()
    xd = A['x'].to(dev).requires_grad_()
    y = m(xd)
    assert maxdiff(y, A['y']) < 2e-5
    y.backward(A['dy'].to(dev))
    e = relerr(xd.grad, A['dx'])
    report('ctc_uni_backward', dx=e)
    assert e < 2e-4
    grads = A['grad']
    n = 0
    for k, p in m.named_parameters():
        if k.endswith('conv.bias') and meta['cfg']['batch_norm']:      # analytically zero (a bias in front of a BatchNorm): fp32 round-off
            assert float(p.grad.abs().max()) < 1e-4 and float(grads[k].abs().max()) < 1e-4, k
        else:
            assert relerr(p.grad, grads[k]) < 2e-4, k
        n += 1
    assert n == len(grads)


@pytest.mark.parametrize('name', ['speech_first_paired', 'speech_first_unpaired'])
def test_speech_first_step_against_reference_golden(dev, name):
    """The speech -> text -> speech training step (bin/train_vqvae.py:159-176,208-270) against what the REAL reference
    produced for the same weights, batch and dropout masks: CTC loss on the paired posteriors, freq_loss on the paired
    (and unpaired) reconstructions, the grad norm and EVERY parameter gradient -- speech encoder, codebook (straight-through
    estimator + p_code path), speaker table and TTS branch."""
    import json
    from argparse import Namespace
    from conftest import load_golden
    from helpers import masks_to, split_masks, tiny_vqvae
    from semi_tts_amd.optim import Optimizer
    from semi_tts_amd.solver import VqvaeTrainer
    W, A, meta = load_golden(name)
    hp, h = meta['hp'], meta['hparas']
    config = dict(data=dict(audio=meta['audio'], corpus=dict(batch_size=3)), hparas=h, model=meta['model'])
    tr = VqvaeTrainer(config, Namespace(vocab_size=meta['vocab_size'], n_spkr=meta['n_spkr'], verbose=False, max_step=1), 'train')
    tr.model = tiny_vqvae(meta, W, dev, strict=True).train()
    tr.optimizer = Optimizer(tr.model.parameters(), h['optimizer'], h['lr'], h['lr_scheduler'], tf_start=h['tf_start'],
                             tf_end=h['tf_end'], tf_step=h['tf_step'])
    text, sid, mel, linear = (A[k].to(dev) for k in ('text', 'sid', 'mel', 'linear'))
    un = {}
    if meta['with_unpaired']:
        un = dict(unpair_mel=A['unpair_mel'].to(dev), unpair_aug_mel=A['unpair_mel'].to(dev), unpair_linear=A['unpair_linear'].to(dev),
                  unpair_sid=A['unpair_sid'].to(dev))
    Bt = text.shape[0]
    B = Bt + (A['unpair_mel'].shape[0] if meta['with_unpaired'] else 0)
    r = hp['n_frames_per_step']
    steps = max(mel.shape[1], A['unpair_mel'].shape[1] if meta['with_unpaired'] else 0) // r
    masks = masks_to(split_masks(A['mask'], hp, True, 1.0, B, B, steps, list(range(steps)), hp['prenet_dim']), dev)
    grads = {}
    orig = tr.clip_grad_norm_

    def spy(params, max_norm):
        for kname, p in tr.model.named_parameters():
            if p.grad is not None:
                grads[kname] = p.grad.detach().clone()
        return orig(list(params), max_norm)
    tr.clip_grad_norm_ = spy
    tr.step = 1          # the unpaired term only counts once step > unpair_speech_start_step (0 in the shipped configs; train_vqvae.py:232)
    st = tr.speech_first_step(mel, mel, linear, text, sid, _masks=masks, **un)
    ref = meta['stats']
    assert torch.equal(tr.model.codebook.last_idx.cpu(), A['idx'])                      # VQ indices: bit-exact
    report('speech_first', name=name, **{k: st[k] for k in ref}, **{'ref_' + k: v for k, v in ref.items()})
    for k in ('asr_loss', 'tts_loss', 'loss') + (('unpair_speech_loss',) if meta['with_unpaired'] else ()):
        assert abs(st[k] - ref[k]) < 2e-5 * max(1.0, abs(ref[k])), (k, st[k], ref[k])
    assert abs(st['grad_norm'] - ref['grad_norm']) < 2e-4 * ref['grad_norm']
    keys = json.loads(bytes(A['grad_keys']).decode())
    worst, worst_k = 0.0, ''
    gmax = max(float(g.abs().max()) for g in A['grad'])
    for k, gref in zip(keys, A['grad']):
        assert k in grads, 'missing gradient for ' + k
        if float(gref.abs().max()) < 1e-6 * gmax:
            # analytically zero (a conv bias in front of a batch-statistics BatchNorm): round-off on both sides
            assert float(grads[k].abs().max()) < 1e-5 * gmax, k
            continue
        e = relerr(grads[k], gref)
        if e > worst:
            worst, worst_k = e, k
        assert e < 1e-3, (k, e)              # fp32 HIP vs fp32 reference, different summation orders on both sides
    report('speech_first_grads', name=name, worst=worst, worst_k=worst_k, n=len(keys))


@pytest.mark.parametrize('B,T,V,L', [(3, 12, 43, 5), (4, 129, 43, 43), (2, 40, 512, 9), (1, 3, 7, 4)])
def test_ctc_loss_against_torch(dev, B, T, V, L):
    """The CTC kernel (alpha / beta recursions, one workgroup per utterance) against torch.nn.CTCLoss() on CPU in float64, the
    way the reference calls it (bin/train_vqvae.py:430-444): log(prob + 1e-10), targets = non-zero tokens (repeated labels,
    zeros in the middle of a row, an empty transcript), all frames, mean over nll / target length."""
    import torch.nn.functional as F
    from semi_tts_amd import autograd as AG
    g = torch.Generator().manual_seed(B * 1000 + T)
    prob = torch.softmax(torch.randn(B, T, V, generator=g) * 2.0, dim=-1)
    text = torch.randint(1, min(V, 6), (B, L), generator=g)              # few distinct labels -> many repeats
    text[0, L // 2] = 0                                                   # a zero inside the row is skipped, not a terminator
    if B > 1:
        text[1, 2:] = 0                                                   # short transcript
    if B > 2:
        text[2] = 0                                                       # empty transcript
    if T < 2 * L:                                                         # keep the alignment feasible for the tiny case
        text[:, 1:] = 0
    pd = prob.to(dev).requires_grad_()
    loss = AG.ctc_loss(pd, text.to(dev), 1e-10)
    (loss * 1.7).backward()
    pr = prob.double().requires_grad_()
    lp = (pr + 1e-10).transpose(0, 1).log()
    ref = F.ctc_loss(lp, text[text != 0], torch.full((B,), T, dtype=torch.long), (text != 0).sum(-1), blank=0, reduction='mean')
    (ref * 1.7).backward()
    e_loss, e_grad = abs(float(loss.detach()) - float(ref.detach())) / max(1.0, abs(float(ref.detach()))), relerr(pd.grad, pr.grad)
    report('ctc_loss', B=B, T=T, V=V, L=L, loss=float(ref), err_loss=e_loss, err_grad=e_grad)
    # torch's own fp32 CPU kernel is 2.3e-4 from its float64 self at (4, 129, 43, 43): log-space recursions over 129 frames,
    # and d/dprob = (p - occupancy) / p amplifies at small p
    assert e_loss < 2e-6 and e_grad < (2e-5 if T < 20 else 1e-3)


def test_ctc_loss_on_log_probabilities_and_invalid_tokens(dev):
    """compute_ctcloss(..., apply_log=False) (bin/train_vqvae.py:430-434: ASRPostnet's log_softmax goes into CTCLoss as it is):
    value and gradient w.r.t. the log-probabilities against torch float64; a token outside [0, V) gives NaN, not a wild read"""
    import torch.nn.functional as F
    from semi_tts_amd import autograd as AG
    B, T, V, L = 3, 40, 43, 9
    g = torch.Generator().manual_seed(21)
    lp0 = torch.log_softmax(torch.randn(B, T, V, generator=g) * 2.0, dim=-1)
    text = torch.randint(1, 6, (B, L), generator=g)
    text[1, 3:] = 0
    pd = lp0.to(dev).requires_grad_()
    loss = AG.ctc_loss(pd, text.to(dev), 1e-10, apply_log=False)
    (loss * 0.7).backward()
    pr = lp0.double().requires_grad_()
    ref = F.ctc_loss(pr.transpose(0, 1), text[text != 0], torch.full((B,), T, dtype=torch.long), (text != 0).sum(-1), blank=0,
                     reduction='mean')
    (ref * 0.7).backward()
    e_loss, e_grad = abs(float(loss.detach()) - float(ref.detach())) / max(1.0, abs(float(ref.detach()))), relerr(pd.grad, pr.grad)
    report('ctc_loss_log_input', err_loss=e_loss, err_grad=e_grad)
    assert e_loss < 2e-6 and e_grad < 2e-4
    bad = text.clone()
    bad[0, 1] = V + 5
    l2 = AG.ctc_loss(lp0.to(dev), bad.to(dev), 1e-10, apply_log=False)
    assert bool(torch.isnan(l2))


def test_asr_postnet_backward_vs_oracle(dev):
    """ASRPostnet with gradients (eval mode: its two dropouts of 0.5 off) against float64 autograd through the oracle, and with
    explicit dropout masks in training mode; the CTC loss on its log-probabilities closes the chain the trainer uses
    (bin/train_vqvae.py:210-213)."""
    from conftest import load_golden
    from oracle import asr_oracle as AO
    from oracle import tts_oracle as O
    from semi_tts_amd.asr import ASRPostnet
    W, A, meta = load_golden('asr_postnet_tiny')
    m = ASRPostnet(meta['latent_dim'], meta['vocab_size'])
    m.load_state_dict(W)
    m = m.to(dev).eval()
    x = A['x']
    xd = x.to(dev).requires_grad_()
    y = m(xd)
    assert maxdiff(y, A['y']) < 2e-5
    dy = rnd(*y.shape, seed=9)
    y.backward(dy.to(dev))
    outs, wg, ig = oracle_grads(lambda Wd, xx: AO.asr_postnet_forward(Wd, xx), W, [x], [dy])
    e = relerr(xd.grad, ig[0])
    report('asr_postnet_backward', dx=e)
    assert e < 2e-4
    check_param_grads(m, '', wg, 2e-4, 'asr_postnet_backward')
    # training mode with explicit (scaled) masks
    m.zero_grad()
    m.train()
    g = torch.Generator().manual_seed(4)
    masks = [torch.bernoulli(torch.full((3, 9, 24), 0.5), generator=g) * 2.0 for _ in range(2)]
    xd2 = x.to(dev).requires_grad_()
    y2 = m(xd2, _masks=[mk.to(dev) for mk in masks])
    y2.backward(dy.to(dev))
    drop = lambda: O.DropoutSource('list', [mk.double() for mk in masks])
    outs2, wg2, ig2 = oracle_grads(lambda Wd, xx: AO.asr_postnet_forward(Wd, xx, training=True, drop=drop()), W, [x], [dy])
    assert maxdiff(y2, outs2[0]) < 2e-5 and relerr(xd2.grad, ig2[0]) < 2e-4
    check_param_grads(m, '', wg2, 2e-4, 'asr_postnet_backward_train')


def test_text_first_step_against_reference_golden(dev):
    """The text -> speech -> text training step (bin/train_vqvae.py:186-205,208-224,234-250) against what the REAL reference produced
    for the same weights, batch and dropout masks: paired text teacher-forced, unpaired text decoded from its own outputs (partial
    teacher), the unpaired prediction detached and quantised next to the paired mel with the table detached for the fake part; CTC on
    both posteriors, freq_loss on the paired reconstruction, the grad norm and EVERY parameter gradient."""
    import json
    from argparse import Namespace
    from conftest import load_golden
    from helpers import coin_source, masks_to, split_masks, tiny_vqvae
    from semi_tts_amd.module import plan_decode
    from semi_tts_amd.optim import Optimizer
    from semi_tts_amd.solver import VqvaeTrainer
    import numpy as np
    W, A, meta = load_golden('text_first_unpaired')
    hp, h = meta['hp'], meta['hparas']
    config = dict(data=dict(audio=meta['audio'], corpus=dict(batch_size=3)), hparas=h, model=meta['model'])
    tr = VqvaeTrainer(config, Namespace(vocab_size=meta['vocab_size'], n_spkr=meta['n_spkr'], verbose=False, max_step=1), 'train')
    tr.model = tiny_vqvae(meta, W, dev, strict=True).train()
    tr.optimizer = Optimizer(tr.model.parameters(), h['optimizer'], h['lr'], h['lr_scheduler'], tf_start=h['tf_start'],
                             tf_end=h['tf_end'], tf_step=h['tf_step'])
    text, sid, mel, linear, utext, usid = (A[k].to(dev) for k in ('text', 'sid', 'mel', 'linear', 'unpair_text', 'unpair_sid'))
    Bt, B = text.shape[0], text.shape[0] + utext.shape[0]
    r = hp['n_frames_per_step']
    umax = int(6.0 * utext.shape[1])
    umax += umax % r                                          # VQVAE.text_to_speech :158-160
    steps, src = plan_decode(False, mel.shape[1], Bt, B, r, 1.0, hp['drop_dec_in'], umax, coin_source(A['coins']))
    masks = masks_to(split_masks(A['mask'], hp, True, 1.0, B, Bt, steps, src, hp['prenet_dim']), dev)
    grads = {}
    orig = tr.clip_grad_norm_

    def spy(params, max_norm):
        for kname, p in tr.model.named_parameters():
            if p.grad is not None:
                grads[kname] = p.grad.detach().clone()
        return orig(list(params), max_norm)
    tr.clip_grad_norm_ = spy
    saved = np.random.rand
    np.random.rand = coin_source(A['coins'])
    try:
        st = tr.text_first_step(mel, mel, linear, text, sid, unpair_text=utext, unpair_sid=usid, _masks=masks)
    finally:
        np.random.rand = saved
    ref = meta['stats']
    report('text_first', **{k: st[k] for k in ref}, **{'ref_' + k: v for k, v in ref.items()})
    assert torch.equal(tr.model.codebook.last_idx.cpu(), A['idx'])                      # VQ indices: bit-exact
    for k in ('asr_loss', 'tts_loss', 'unpair_text_loss', 'loss'):
        assert abs(st[k] - ref[k]) < 2e-5 * max(1.0, abs(ref[k])), (k, st[k], ref[k])
    assert abs(st['grad_norm'] - ref['grad_norm']) < 2e-4 * ref['grad_norm']
    keys = json.loads(bytes(A['grad_keys']).decode())
    worst, worst_k = 0.0, ''
    gmax = max(float(g.abs().max()) for g in A['grad'])
    for k, gref in zip(keys, A['grad']):
        assert k in grads, 'missing gradient for ' + k
        if float(gref.abs().max()) < 1e-6 * gmax:
            assert float(grads[k].abs().max()) < 1e-5 * gmax, k
            continue
        e = relerr(grads[k], gref)
        if e > worst:
            worst, worst_k = e, k
        assert e < 1e-3, (k, e)
    report('text_first_grads', worst=worst, worst_k=worst_k, n=len(keys))


def test_bptt_loop_with_hosted_attention_backward_is_bitwise_the_plain_loop(dev):
    """The software-pipelined BPTT loop (decoder cell product of step t-1 in ONE launch with the attention backward of step t,
    st_skinny_linear_packed_lstm_bwd_attn_bwd) only re-schedules the same kernels' work: every gradient of the full-size decoder
    (B = 32, L = 43, 6 steps, dropout on) is bit-identical to the loop with separate launches, and to the six-launch form's within
    round-off.  The split form (bwd_attn_parts = 2: two attention workgroups per utterance over halves of the attention dims, the rest of
    the step's attention backward in the W_q^T dpq launch) agrees within round-off and reproduces itself bit for bit."""
    from helpers import full_tacotron
    m = full_tacotron(dev, seed=4321, prenet_dropout=0.5).train()
    dec = m.decoder
    B, L, steps = 32, 43, 6
    r, n_mels = dec.n_frames_per_step, dec.n_mels
    mem0, spk0 = rnd(B, L, 512, seed=1).to(dev), rnd(B, 128, seed=2).to(dev)
    teacher = torch.rand(B, steps * r, n_mels, generator=torch.Generator().manual_seed(3)).to(dev)
    douts = None
    res = {}
    for mode in ('overlap', 'split', 'plain', 'six'):
        dec.bwd_overlap_attn = mode in ('overlap', 'split')
        dec.bwd_attn_parts = 2 if mode == 'split' else 1        # 'split': the hosted attention backward as two workgroups per utterance
        dec.bwd_fuse_pointwise = mode != 'six'
        dec.fwd_pair_cells = mode in ('overlap', 'split')  # (the forward's paired LSTM cells too: same work, one launch for two cells)
        for p in dec.parameters():
            p.grad = None
        torch.manual_seed(77)                                   # the same dropout masks in every pass
        mem, spk = mem0.clone().requires_grad_(), spk0.clone().requires_grad_()
        mel, align, stop = dec(mem, None, teacher, spk, tf_rate=1.0)
        if douts is None:
            douts = [rnd(*mel.shape, seed=5).to(dev), rnd(*align.shape, seed=6).to(dev), rnd(*stop.shape, seed=7).to(dev)]
        torch.autograd.backward([mel, align, stop], douts)
        res[mode] = dict(mel=mel.detach().clone(), dmem=mem.grad.clone(), dspk=spk.grad.clone(),
                         **{k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None})
    dec.bwd_overlap_attn = dec.bwd_fuse_pointwise = dec.fwd_pair_cells = True
    dec.bwd_attn_parts = 2
    assert len(res['overlap']) > 20
    for k, v in res['overlap'].items():
        assert torch.equal(v, res['plain'][k]), k
        assert relerr(v, res['six'][k]) < 1e-5, k
        # the split form adds the two halves' location-feature gradients after the products instead of inside them: round-off only
        assert relerr(res['split'][k], v) < 1e-5, k
    # ... and it is what a second run reproduces bit for bit (fixed summation order, no atomics)
    dec.bwd_attn_parts = 2
    for p in dec.parameters():
        p.grad = None
    torch.manual_seed(77)
    mem, spk = mem0.clone().requires_grad_(), spk0.clone().requires_grad_()
    mel, align, stop = dec(mem, None, teacher, spk, tf_rate=1.0)
    torch.autograd.backward([mel, align, stop], douts)
    again = dict(dmem=mem.grad.clone(), **{k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None})
    for k, v in again.items():
        assert torch.equal(v, res['split'][k]), k


@pytest.mark.parametrize('B,L', [(20, 43), (32, 61), (16, 43), (33, 20), (24, 97)])
def test_split_bptt_forms_agree_across_shapes(dev, B, L):
    """The split forms of the BPTT loop (two attention workgroups per utterance; K-split partial product of the decoder cell when 16 < B <= 32
    is a multiple of 16; history part in the dgates_q launch) against the whole-step-per-workgroup loop on shapes around their limits: pad
    rows (B = 20, 24, 33: no partial product, zero-filled tapes), one batch tile (B = 16), longer texts (L = 61, 97: several position
    blocks).  Same arithmetic up to the order in which the two halves' location-feature gradients meet."""
    from helpers import full_tacotron
    m = full_tacotron(dev, seed=777, prenet_dropout=0.5).train()
    dec = m.decoder
    steps = 4
    r, n_mels = dec.n_frames_per_step, dec.n_mels
    mem0, spk0 = rnd(B, L, 512, seed=1).to(dev), rnd(B, 128, seed=2).to(dev)
    teacher = torch.rand(B, steps * r, n_mels, generator=torch.Generator().manual_seed(3)).to(dev)
    res = {}
    for parts in (1, 2):
        dec.bwd_attn_parts = parts
        for p in dec.parameters():
            p.grad = None
        torch.manual_seed(11)
        mem, spk = mem0.clone().requires_grad_(), spk0.clone().requires_grad_()
        mel, align, stop = dec(mem, None, teacher, spk, tf_rate=1.0)
        torch.autograd.backward([mel, align, stop], [torch.ones_like(mel), rnd(*align.shape, seed=6).to(dev), torch.ones_like(stop)])
        res[parts] = dict(dmem=mem.grad.clone(), dspk=spk.grad.clone(), **{k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None})
    dec.bwd_attn_parts = 2
    assert len(res[2]) > 20
    for k, v in res[2].items():
        assert torch.isfinite(v).all(), k
        assert relerr(v, res[1][k]) < 1e-5, k


def test_async_training_step_equals_the_synchronous_one_and_skips_nan_steps_on_device(dev):
    """TtsTrainer.async_stats: no host read inside the step (LazyStats; Adam guarded by the device-side gradient norm).  Two steps give
    bit-identical weights and statistics to the synchronous trainer; a step whose gradient norm is NaN leaves weights and Adam moments
    untouched, as BaseSolver.backward's `if math.isnan(grad_norm)` does (src/solver.py:147-150)."""
    from argparse import Namespace
    from conftest import load_golden
    from helpers import masks_to, split_masks, tiny_vqvae
    from semi_tts_amd.optim import Optimizer
    from semi_tts_amd.solver import LazyStats, TtsTrainer
    W, A, meta = load_golden('train_step_tiny')
    hp, h = meta['hp'], meta['hparas']
    config = dict(data=dict(audio=meta['audio'], corpus=dict(batch_size=3)), hparas=h, model=meta['model'])
    text, sid, mel, linear = (A[k].to(dev) for k in ('text', 'sid', 'mel', 'linear'))
    B, steps = text.shape[0], mel.shape[1] // hp['n_frames_per_step']

    def run(async_stats, poison=False):
        tr = TtsTrainer(config, Namespace(vocab_size=meta['vocab_size'], n_spkr=meta['n_spkr'], verbose=False, max_step=2), 'train')
        tr.model = tiny_vqvae(meta, W, dev).train()
        tr.optimizer = Optimizer(tr.model.parameters(), h['optimizer'], h['lr'], h['lr_scheduler'], tf_start=h['tf_start'],
                                 tf_end=h['tf_end'], tf_step=h['tf_step'])
        tr.async_stats = async_stats
        out, pos = [], 0
        for n in meta['n_masks']:
            masks = masks_to(split_masks(A['mask'][pos:pos + n], hp, True, 1.0, B, B, steps, list(range(steps)), hp['prenet_dim']), dev)
            pos += n
            m = mel.clone()
            if poison:
                m[0, 0, 0] = float('nan')
            out.append(tr.train_step(text, sid, m, linear, _masks=masks))
        return tr, out
    tr_s, st_s = run(False)
    tr_a, st_a = run(True)
    assert isinstance(st_a[0], LazyStats) and torch.is_tensor(dict.__getitem__(st_a[0], 'loss'))      # nothing was read inside the step
    for a, s in zip(st_a, st_s):
        assert a['loss'] == s['loss'] and a['grad_norm'] == s['grad_norm'] and a['lr'] == s['lr']
    trained = [k for k, p in tr_s.model.named_parameters() if p.grad is not None]     # (the ASR branch is not part of this step and is
    assert len(trained) > 50                                                           #  initialised randomly per construction)
    sd_a, sd_s = dict(tr_a.model.named_parameters()), dict(tr_s.model.named_parameters())
    for k in trained:
        assert torch.equal(sd_a[k].detach(), sd_s[k].detach()), k
    tr_n, st_n = run(True, poison=True)
    assert st_n[0]['grad_norm'] != st_n[0]['grad_norm']                       # NaN gradient norm ...
    w0 = {k: v.to(dev) for k, v in tiny_vqvae(meta, W, dev).state_dict().items()}
    for k, p in tr_n.model.named_parameters():                                # ... and no parameter moved
        if k in W:
            assert torch.equal(p.detach(), w0[k]), k
    for st in tr_n.optimizer.opt.state.values():
        assert float(st['exp_avg'].abs().max()) == 0.0 and float(st['exp_avg_sq'].abs().max()) == 0.0
    # reading the norms rolled the host-side Adam step counts of exactly the parameters of those updates back (bias corrections and the
    # checkpointed `step` then agree with an optimiser that never saw the skipped steps); every unread step is read here
    tr_n.drain_stats()
    assert all(not math.isfinite(s['grad_norm']) for s in st_n)
    assert all(st.get('_step', 0) == 0 for st in tr_n.optimizer.opt.state.values())
    assert tr_n.optimizer.opt.guarded_steps == len(st_n) and not tr_n.optimizer.opt.__dict__.get('_guarded_log')


@pytest.mark.parametrize('M,N,Cin,KT', [(2752, 4096, 160, 1), (300, 70, 33, 1), (8256, 128, 80, 3), (5000, 200, 8, 5), (64, 16, 16, 1)])
def test_weight_gradient_with_bias_gradient_in_the_same_launches(dev, M, N, Cin, KT):
    """st_gemm_wgrad_db: dW as st_gemm_wgrad gives it (bit for bit) and db = column sums of dC, for one and many row slabs, both tile
    sizes, the folded few-channel form and widths that are not multiples of the tile."""
    from semi_tts_amd import ops
    Bn = 1 if KT == 1 else 4
    T = M // Bn
    a = rnd(Bn, T, Cin, seed=1).to(dev)
    dc = rnd(Bn, T, N, seed=2).to(dev)
    pad = KT // 2
    dw_ref = ops.gemm_wgrad(dc, a, KT, pad)
    dw, db = ops.gemm_wgrad(dc, a, KT, pad, with_db=True)
    assert torch.equal(dw, dw_ref)
    ref = dc.cpu().double().sum((0, 1))
    assert relerr(db, ref) < 2e-6


def test_hosted_attention_backward_falls_back_for_texts_that_fill_the_lds(dev):
    """L = 330: the attention backward alone needs 158 KB of LDS, so it cannot share a workgroup's LDS budget with the product it is hosted
    beside (st_skinny_linear_packed_lstm_bwd_attn_bwd then runs the two launches one after the other): same gradients as the plain loop."""
    from helpers import full_tacotron
    m = full_tacotron(dev, seed=99, prenet_dropout=0.5).train()
    dec = m.decoder
    B, L, steps = 3, 330, 3
    r, n_mels = dec.n_frames_per_step, dec.n_mels
    mem0, spk0 = rnd(B, L, 512, seed=1).to(dev), rnd(B, 128, seed=2).to(dev)
    teacher = torch.rand(B, steps * r, n_mels, generator=torch.Generator().manual_seed(3)).to(dev)
    res = {}
    for mode in (True, False):
        dec.bwd_overlap_attn = mode
        for p in dec.parameters():
            p.grad = None
        torch.manual_seed(5)
        mem, spk = mem0.clone().requires_grad_(), spk0.clone().requires_grad_()
        mel, align, stop = dec(mem, None, teacher, spk, tf_rate=1.0)
        torch.autograd.backward([mel, align, stop], [torch.ones_like(mel), torch.ones_like(align), torch.ones_like(stop)])
        res[mode] = dict(dmem=mem.grad.clone(), **{k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None})
    dec.bwd_overlap_attn = True
    for k, v in res[True].items():
        assert torch.isfinite(v).all() and torch.equal(v, res[False][k]), k
