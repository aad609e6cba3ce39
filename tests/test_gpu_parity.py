"""Parity of the HIP path (through the C ABI) against the CPU oracle and the golden vectors
recorded from the real reference.  Needs a real MI355X:  pytest -m gpu

Tolerances: the north star asks for 1e-3 max-abs on mel (fp32) and bit-exact VQ indices; the
kernels use exact-fp32 MFMA so the observed error is summation-order noise (~1e-6) and the
tests assert much tighter bounds, written next to each check.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import (coin_source, full_hp, full_tacotron, masks_to, maxdiff, report, split_masks, tiny_tacotron)

pytestmark = pytest.mark.gpu

from oracle import tts_oracle as O   # noqa: E402
from oracle import vq_oracle as VQ   # noqa: E402


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'the gpu-marked tests need a GPU'
    from semi_tts_amd import _lib
    _lib.load()     # fail loudly if the HIP library is missing
    return torch.device('cuda:0')


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


# ------------------------------------------------------------------------------------ op level
@pytest.mark.parametrize('B,N,K', [(1, 16, 32), (4, 40, 24), (16, 256, 240), (17, 241, 1536), (32, 256, 1024),
                                   (33, 48, 100), (64, 80, 62), (70, 33, 31)])
def test_skinny_linear(dev, B, N, K):
    from semi_tts_amd import ops
    x, w, b = rnd(B, K, seed=1), rnd(N, K, scale=K ** -0.5, seed=2), rnd(N, seed=3)
    mask = (torch.rand(B, N) > 0.5).float() * 2
    y = ops.linear_small(x.to(dev), w.to(dev), b.to(dev), 'relu', mask.to(dev))
    ref = torch.relu(x.double() @ w.double().t() + b.double()) * mask.double()
    err = maxdiff(y, ref)
    report('skinny_linear', B=B, N=N, K=K, err=err)
    assert err < 2e-5     # fp32 accumulation over K <= 1536 of O(1) terms


def test_skinny_linear_split_output(dev):
    # proj (+) gate in one launch: columns >= n_split go to y2, repeated r times (module.py:285-287)
    from semi_tts_amd import ops
    B, K1, K2, N, r = 5, 40, 32, 25, 3
    x1, x2 = rnd(B, K1, seed=4), rnd(B, K2, seed=5)
    w, b = rnd(N, K1 + K2, scale=0.1, seed=6).to(dev), rnd(N, seed=7)
    x1d, x2d = x1.to(dev), x2.to(dev)
    y = torch.zeros(B, 2, N - 1, device=dev)      # strided output: row stride 2*(N-1)
    y2 = torch.zeros(B, 2 * r, device=dev)
    ops.skinny_linear([ops.seg(x1d, w, k=K1, ldw=K1 + K2), ops.seg(x2d, w[:, K1:], k=K2, ldw=K1 + K2)],
                      y[:, 1], bias=b.to(dev), n_split=N - 1, y2=y2[:, r:], rep=r, N=N)
    ref = torch.cat([x1, x2], 1).double() @ w.cpu().double().t() + b.double()
    assert maxdiff(y[:, 1], ref[:, :N - 1]) < 1e-5
    assert maxdiff(y2[:, r:], ref[:, N - 1:].repeat(1, r)) < 1e-5
    assert float(y[:, 0].abs().max()) == 0 and float(y2[:, :r].abs().max()) == 0


@pytest.mark.parametrize('B,H,K1,K2', [(2, 8, 12, 20), (4, 48, 16, 32), (32, 1024, 256, 512), (64, 256, 64, 0), (19, 40, 30, 8)])
def test_lstm_cell(dev, B, H, K1, K2):
    from semi_tts_amd import ops
    K = K1 + K2
    x, h, c = rnd(B, K, seed=1), rnd(B, H, seed=2), rnd(B, H, seed=3)
    w_ih, w_hh = rnd(4 * H, K, scale=K ** -0.5, seed=4), rnd(4 * H, H, scale=H ** -0.5, seed=5)
    b_ih, b_hh = rnd(4 * H, scale=0.1, seed=6), rnd(4 * H, scale=0.1, seed=7)
    mask = (torch.rand(B, H) > 0.1).float() / 0.9
    h_ref, c_ref = O.lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh)
    xd, hd, w_ihd, w_hhd = x.to(dev), h.to(dev), w_ih.to(dev), w_hh.to(dev)
    segs = [ops.seg(xd, w_ihd, k=K1, ldx=K, ldw=K)]
    if K2:
        segs.append(ops.seg(xd[:, K1:], w_ihd[:, K1:], k=K2, ldx=K, ldw=K))
    segs.append(ops.seg(hd, w_hhd))
    h_out, c_out = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
    gates = torch.empty(B, 4, H, device=dev)
    ops.lstm_cell(segs, b_ih.to(dev), b_hh.to(dev), c.to(dev), h_out, c_out, mask=mask.to(dev), gates_out=gates)
    e_h, e_c = maxdiff(h_out, h_ref * mask), maxdiff(c_out, c_ref)
    report('lstm_cell', B=B, H=H, K=K, err_h=e_h, err_c=e_c)
    assert e_h < 1e-5 and e_c < 1e-5
    g_ref = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    assert maxdiff(gates[:, 2], torch.tanh(g_ref[:, 2 * H:3 * H])) < 1e-5
    assert maxdiff(gates[:, 0], torch.sigmoid(g_ref[:, :H])) < 1e-5


@pytest.mark.parametrize('B,K', [(1, 4), (5, 24), (16, 16), (17, 33), (32, 240), (70, 100)])
def test_tile_untile_roundtrip(dev, B, K):
    from semi_tts_amd import ops
    x = rnd(B, K, seed=3).to(dev)
    t = ops.tile_rows(x)
    assert t.numel() == ((B + 15) // 16) * ((K + 15) // 16) * 256
    assert torch.equal(ops.untile_rows(t, B, K), x)
    # pads are zero and the logical elements are a permutation of x
    assert abs(float(t.abs().sum()) - float(x.abs().sum())) < 1e-3 * (1 + float(x.abs().sum()))


def _combined_t16(ops, xs, Ks, dev):
    """tile several (B, k_s) activations into ONE T16 buffer, each at its own k-block range"""
    B = xs[0].shape[0]
    kbs = [ops.kb16(k) for k in Ks]
    stride = sum(kbs)
    buf = torch.zeros(((B + 15) // 16) * stride * 256, device=dev)
    kb0 = 0
    for x, k, n in zip(xs, Ks, kbs):
        ops.tile_rows(x.to(dev), out=buf, kb_stride=stride, kb0=kb0)
        kb0 += n
    return buf, stride


@pytest.mark.parametrize('B,H,Ks', [(2, 8, (12, 20, 8)), (4, 48, (16, 32, 48)), (32, 1024, (256, 512, 1024)),
                                    (64, 256, (64, 256)), (19, 40, (30, 40)), (70, 16, (24,)), (33, 64, (48, 64))])
def test_lstm_cell_packed(dev, B, H, Ks):
    """the decode loop's LSTM cell on pre-packed (P16) weights and ONE tiled (T16) activation run"""
    from semi_tts_amd import ops
    K = sum(Ks)
    xs = [rnd(B, k, seed=10 + i) for i, k in enumerate(Ks)]
    c = rnd(B, H, seed=3)
    w = rnd(4 * H, K, scale=K ** -0.5, seed=4)
    b_ih, b_hh = rnd(4 * H, scale=0.1, seed=6), rnd(4 * H, scale=0.1, seed=7)
    mask = (torch.rand(B, H) > 0.1).float() / 0.9
    std, mean = torch.relu(rnd(B, H, seed=8)), rnd(B, H, seed=9)
    gates = torch.cat(xs, 1) @ w.t() + b_ih + b_hh
    i, f, g, o = [gates[:, j * H:(j + 1) * H] for j in range(4)]
    c_ref = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h_ref = torch.sigmoid(o) * torch.tanh(c_ref) * mask
    wd = w.to(dev)
    offs = np.cumsum([0] + list(Ks))
    packed = ops.pack_weight([wd[:, offs[j]:] for j in range(len(Ks))], Ks, 4 * H, lstm_H=H, ldws=[K] * len(Ks))
    xbuf, stride = _combined_t16(ops, xs, Ks, dev)
    # the new h goes to two destinations: its own buffer and a slice of a wider one
    h_t16 = torch.zeros(ops.t16_floats(B, H), device=dev)
    wide_kbs = ops.kb16(H) + 3
    wide = torch.zeros(((B + 15) // 16) * wide_kbs * 256, device=dev)
    ha_t16 = torch.zeros(ops.t16_floats(B, H), device=dev)
    c_out = torch.empty(B, H, device=dev)
    gts = torch.empty(B, 4, H, device=dev)
    ops.lstm_cell_packed(packed, ops.t16_view(xbuf, stride), 16 * stride, b_ih.to(dev), b_hh.to(dev), c.to(dev),
                         ops.t16_view(h_t16, K=H), c_out, B, H, h_dst1=ops.t16_view(wide, wide_kbs, 2), mask=mask.to(dev),
                         gates_out=gts, ada_std=std.to(dev), ada_mean=mean.to(dev), hadapt_dst=ops.t16_view(ha_t16, K=H))
    h = ops.untile_rows(h_t16, B, H)
    e_h, e_c = maxdiff(h, h_ref), maxdiff(c_out, c_ref)
    report('lstm_cell_packed', B=B, H=H, K=K, err_h=e_h, err_c=e_c)
    assert e_h < 1e-5 and e_c < 1e-5
    assert torch.equal(ops.untile_rows(wide, B, H, kb_stride=wide_kbs, kb0=2), h)
    assert maxdiff(ops.untile_rows(ha_t16, B, H), std * (h_ref - mean)) < 1e-5
    assert maxdiff(gts[:, 3], torch.sigmoid(o)) < 1e-5


@pytest.mark.parametrize('K,cols', [(64, (32,)), (4096, (256, 512, 1024)), (100, (24, 40)), (37, (5, 18, 9)), (128, (1024,)), (48, (30, 34))])
def test_pack_weight_transposed_concat(dev, K, cols):
    """st_pack_weight_t: the P16 image of cat(ws, 1).t() straight from the parameters == st_pack_weight of the materialised
    transpose, bit for bit (aligned and ragged segment boundaries, K and N that are not multiples of 16, strided sources)."""
    from semi_tts_amd import ops
    big = rnd(K, sum(cols) + 3, seed=5).to(dev)
    ws, off = [], 0
    for i, c in enumerate(cols):
        ws.append(big[:, off:off + c] if i % 2 == 0 else big[:, off:off + c].contiguous())      # views with a row stride > cols too
        off += c
    ref = ops.pack_weight([torch.cat(ws, 1).t().contiguous()], [K], sum(cols))
    got = ops.pack_weight_t(ws)
    assert got.shape == ref.shape and torch.equal(got, ref)


@pytest.mark.parametrize('B,N,Ks', [(1, 16, (32,)), (4, 40, (24,)), (32, 241, (1024, 512)), (32, 256, (240,)),
                                    (33, 48, (100, 28)), (64, 80, (62,)), (70, 33, (31,))])
def test_skinny_linear_packed(dev, B, N, Ks):
    from semi_tts_amd import ops
    K = sum(Ks)
    xs = [rnd(B, k, seed=20 + i) for i, k in enumerate(Ks)]
    w, b = rnd(N, K, scale=K ** -0.5, seed=2), rnd(N, seed=3)
    mask = (torch.rand(B, N) > 0.5).float() * 2
    ref = torch.relu(torch.cat(xs, 1).double() @ w.double().t() + b.double()) * mask.double()
    wd = w.to(dev)
    offs = np.cumsum([0] + list(Ks))
    packed = ops.pack_weight([wd[:, offs[j]:] for j in range(len(Ks))], Ks, N, ldws=[K] * len(Ks))
    xbuf, stride = _combined_t16(ops, xs, Ks, dev)
    xv = ops.t16_view(xbuf, stride)
    y = torch.zeros(B, N, device=dev)
    y_t16 = torch.zeros(ops.t16_floats(B, N), device=dev)
    ops.skinny_linear_packed(packed, xv, 16 * stride, B, N, y=y, y_dst=ops.t16_view(y_t16, K=N), bias=b.to(dev),
                             act='relu', mask=mask.to(dev))
    err = maxdiff(y, ref)
    report('skinny_linear_packed', B=B, N=N, K=K, err=err)
    assert err < 2e-5
    assert torch.equal(ops.untile_rows(y_t16, B, N), y)
    # split output (proj + gate): the last column goes to y2, repeated
    if N > 16:
        y1 = torch.zeros(B, N - 1, device=dev)
        y2 = torch.zeros(B, 3, device=dev)
        t1 = torch.zeros(ops.t16_floats(B, N - 1), device=dev)
        ops.skinny_linear_packed(packed, xv, 16 * stride, B, N, y=y1, y_dst=ops.t16_view(t1, K=N - 1), bias=b.to(dev),
                                 n_split=N - 1, y2=y2, rep=3)
        lin = torch.cat(xs, 1).double() @ w.double().t() + b.double()
        assert maxdiff(y1, lin[:, :N - 1]) < 2e-5 and maxdiff(y2, lin[:, N - 1:].repeat(1, 3)) < 2e-5
        assert torch.equal(ops.untile_rows(t1, B, N - 1), y1)


@pytest.mark.parametrize('B,L,A,E,F,K,Q', [(2, 7, 16, 32, 4, 5, 48), (4, 12, 256, 512, 32, 31, 1024),
                                         (32, 43, 256, 512, 32, 31, 1024), (3, 171, 256, 512, 32, 31, 1024),
                                         (2, 70, 20, 24, 3, 3, 10)])
def test_attention_step(dev, B, L, A, E, F, K, Q):
    from semi_tts_amd import ops
    W = {'p.query_layer.linear.weight': rnd(A, Q, scale=Q ** -0.5, seed=1),
         'p.loc_conv.conv.weight': rnd(F, 2, K, scale=0.3, seed=2),
         'p.loc_linear.linear.weight': rnd(A, F, scale=0.3, seed=3),
         'p.v.linear.weight': rnd(1, A, scale=0.5, seed=4)}
    query, memory, pm = rnd(B, Q, seed=5), rnd(B, L, E, seed=6), rnd(B, L, A, seed=7)
    w_prev = torch.softmax(rnd(B, L, seed=8), -1)
    w_cum = w_prev + torch.softmax(rnd(B, L, seed=9), -1)
    ctx_ref, w_ref = O.attention_step(W, query, memory, pm, w_prev, w_cum, 'p.')
    std, mean = torch.relu(rnd(B, Q, seed=10)), rnd(B, Q, seed=11)
    d = lambda t: t.to(dev).contiguous()
    pq = ops.linear_small(d(query), d(W['p.query_layer.linear.weight']))
    w_out, w_cum_out = torch.empty(B, L, device=dev), torch.empty(B, L, device=dev)
    ctx, hadapt = torch.empty(B, E, device=dev), torch.empty(B, Q, device=dev)
    ops.attn_step(pq, d(pm), d(memory), d(w_prev), d(w_cum), w_out, w_cum_out, d(W['p.loc_conv.conv.weight']),
                  d(W['p.loc_linear.linear.weight']), d(W['p.v.linear.weight']), ctx,
                  h_q=d(query), ada_std=d(std), ada_mean=d(mean), h_adapt=hadapt)
    e_w, e_c = maxdiff(w_out, w_ref), maxdiff(ctx, ctx_ref)
    report('attn_step', B=B, L=L, A=A, err_w=e_w, err_ctx=e_c)
    assert e_w < 2e-6 and e_c < 2e-5
    assert maxdiff(w_cum_out, w_ref + w_cum) < 2e-6
    assert maxdiff(hadapt, std * (query - mean)) < 1e-6
    assert abs(float(w_out.sum(-1).mean()) - 1.0) < 1e-5


@pytest.mark.parametrize('B,L,A,E,F,K', [(2, 7, 16, 32, 4, 5), (32, 43, 256, 512, 32, 31), (3, 171, 256, 512, 32, 31), (5, 13, 24, 40, 6, 7),
                                         (2, 300, 256, 512, 32, 31), (1, 350, 24, 40, 6, 7)])
def test_attention_step_in_two_parts(dev, B, L, A, E, F, K):
    # pre (conv + W_l, from the previous weights only) then fin (energies, softmax, context) == the one-launch step
    from semi_tts_amd import ops
    pq, pm, mem = rnd(B, A, seed=1), rnd(B, L, A, seed=2), rnd(B, L, E, seed=3)
    w_prev = torch.softmax(rnd(B, L, seed=4), -1)
    w_cum = w_prev * 2.5
    wc, wl, v = rnd(F, 2, K, scale=0.3, seed=5), rnd(A, F, scale=0.3, seed=6), rnd(1, A, seed=7)
    d = [t.to(dev) for t in (pq, pm, mem, w_prev, w_cum, wc, wl, v)]
    w1, c1, x1 = (torch.empty(B, L, device=dev), torch.empty(B, L, device=dev), torch.empty(B, E, device=dev))
    ops.attn_step(d[0], d[1], d[2], d[3], d[4], w1, c1, d[5], d[6], d[7], x1)
    s_buf = ops.attn_pre(d[1], d[3], d[4], d[5], d[6])
    for parts in (2, 4, 8, 16):       # the pre part spread over several workgroups per utterance gives the same S bit for bit
        assert torch.equal(ops.attn_pre(d[1], d[3], d[4], d[5], d[6], parts=parts), s_buf)
    w2, c2, x2 = torch.empty_like(w1), torch.empty_like(c1), torch.empty_like(x1)
    ops.attn_fin(d[0], s_buf, d[2], d[4], d[7], w2, c2, x2, F, K)
    for parts in (2, 4, 8):    # the fin part over several workgroups per utterance (slices of the context dims)
        w3, c3, x3 = torch.empty_like(w1), torch.empty_like(c1), torch.empty_like(x1)
        if E % (4 * parts):    # a slice must be whole float4 columns: refused, not mis-computed
            with pytest.raises(RuntimeError):
                ops.attn_fin(d[0], s_buf, d[2], d[4], d[7], w3, c3, x3, F, K, parts=parts)
            continue
        ops.attn_fin(d[0], s_buf, d[2], d[4], d[7], w3, c3, x3, F, K, parts=parts)
        assert torch.equal(w3, w2) and torch.equal(c3, c2)
        assert float((x3 - x2).abs().max()) < 1e-6
    errs = dict(w=maxdiff(w2, w1), cum=maxdiff(c2, c1), ctx=maxdiff(x2, x1))
    report('attention_two_parts', B=B, L=L, A=A, **errs)
    assert errs['w'] < 2e-6 and errs['cum'] < 2e-6 and errs['ctx'] < 2e-5      # (pq + ploc) + pm vs pq + (pm + ploc)


@pytest.mark.parametrize('B,L,A,E,parts', [(3, 171, 256, 512, 4), (2, 300, 256, 512, 7), (5, 40, 24, 64, 2), (1, 900, 256, 512, 21),
                                           (4, 130, 64, 128, 64)])
def test_attention_fin_part_over_position_ranges(dev, B, L, A, E, parts):
    """st_attn_fin_split_fwd (long texts): local softmax statistics and partial contexts per position range + a combine launch
    == the one-workgroup fin part (same softmax, rounded differently) and == softmax(v . tanh(pq + S)) @ memory in float64."""
    from semi_tts_amd import ops
    pq, S, mem = rnd(B, A, seed=1), rnd(B, L, A, seed=2), rnd(B, L, E, seed=3)
    w_cum, v = torch.rand(B, L, generator=torch.Generator().manual_seed(4)), rnd(1, A, seed=7)
    d = [t.to(dev) for t in (pq, S, mem, w_cum, v)]
    w1, c1, x1 = (torch.empty(B, L, device=dev), torch.empty(B, L, device=dev), torch.empty(B, E, device=dev))
    ops.attn_fin(d[0], d[1], d[2], d[3], d[4], w1, c1, x1, 32, 31)
    w2, c2, x2 = torch.empty_like(w1), torch.empty_like(c1), torch.empty_like(x1)
    ops.attn_fin_split(d[0], d[1], d[2], d[3], d[4], w2, c2, x2, parts)
    e = torch.tanh(pq.double().unsqueeze(1) + S.double()).matmul(v.double().view(-1))
    w_r = torch.softmax(e, -1)
    x_r = torch.bmm(w_r.unsqueeze(1), mem.double()).squeeze(1)
    errs = dict(w_vs_fin=maxdiff(w2, w1), ctx_vs_fin=maxdiff(x2, x1), w=maxdiff(w2, w_r), ctx=maxdiff(x2, x_r), cum=maxdiff(c2, w_r + w_cum.double()))
    report('attention_fin_split', B=B, L=L, parts=parts, **errs)
    assert errs['w_vs_fin'] < 1e-6 and errs['ctx_vs_fin'] < 1e-5
    assert errs['w'] < 2e-6 and errs['ctx'] < 2e-5 and errs['cum'] < 2e-6
    assert abs(float(w2.sum(-1).mean()) - 1.0) < 1e-5
    with pytest.raises(RuntimeError):
        ops.attn_fin_split(d[0], d[1], d[2], d[3], d[4], w2, c2, x2, 1)          # parts must be 2..64


@pytest.mark.parametrize('B,L,Q,A,E,parts', [(32, 43, 1024, 256, 512, 2), (5, 9, 48, 16, 32, 1), (17, 30, 64, 32, 64, 2)])
def test_query_projection_and_fin_part_in_one_launch(dev, B, L, Q, A, E, parts):
    """st_query_attn_fin_fwd: pq = W_q h_q handed to the fin workgroups INSIDE the launch as {value, tag} words == the packed
    linear followed by the fin part, bit for bit, for consecutive epochs on the same granule buffer; a launch that would not fit the
    device at once is refused."""
    from semi_tts_amd import ops
    wq, hq = rnd(A, Q, scale=Q ** -0.5, seed=1), rnd(B, Q, seed=2)
    S, mem = rnd(B, L, A, seed=3), rnd(B, L, E, seed=4)
    w_cum, v = torch.rand(B, L, generator=torch.Generator().manual_seed(5)), rnd(1, A, seed=6)
    d = [t.to(dev) for t in (wq, hq, S, mem, w_cum, v)]
    packed = ops.pack_weight([d[0]], [Q], A)
    h_t = ops.tile_rows(d[1])
    pq = torch.empty(B, A, device=dev)                          # the two-launch form: packed linear, then the fin part
    ops.skinny_linear_packed(packed, ops.t16_view(h_t, K=Q), 16 * ops.kb16(Q), B, A, y=pq)
    assert maxdiff(pq, hq.double() @ wq.double().t()) < 1e-5
    w1, c1, x1 = (torch.empty(B, L, device=dev), torch.empty(B, L, device=dev), torch.empty(B, E, device=dev))
    ops.attn_fin(pq, d[2], d[3], d[4], d[5], w1, c1, x1, 32, 31, parts=parts)
    gran = None
    for epoch in (1, 2, 3):
        w2, c2 = torch.empty_like(w1), torch.empty_like(c1)
        x_t = torch.zeros(ops.t16_floats(B, E), device=dev)
        gran = ops.query_attn_fin(packed, h_t, Q, d[2], d[3], d[4], d[5], w2, c2, x_t, 32, 31, parts=parts, epoch=epoch, granules=gran)
        x2 = ops.untile_rows(x_t, B, E)
        errs = dict(w=maxdiff(w2, w1), cum=maxdiff(c2, c1), ctx=maxdiff(x2, x1))
        report('query_attn_fin', B=B, L=L, epoch=epoch, **errs)
        assert errs['w'] == 0.0 and errs['cum'] == 0.0 and errs['ctx'] == 0.0     # the same arithmetic as the two launches
        assert bool(torch.isfinite(x2).all())
    with pytest.raises(RuntimeError):       # epoch 0 is the "never written" tag
        ops.query_attn_fin(packed, h_t, Q, d[2], d[3], d[4], d[5], w2, c2, x_t, 32, 31, parts=parts, epoch=0, granules=gran)


CONV_CASES = [  # B, T, Cin, N, KT, pad, Tout(None = natural)
    (2, 9, 12, 32, 5, 2, None), (3, 13, 80, 80, 4, 2, 13), (3, 13, 80, 80, 4, 2, 14), (2, 20, 640, 128, 3, 1, None),
    (4, 12, 64, 512, 5, 2, None), (2, 7, 8, 8, 8, 4, 7), (1, 130, 160, 1025, 1, 0, None), (2, 5, 30, 70, 1, 0, None),
    (2, 11, 3, 5, 2, 1, 11)]


@pytest.mark.parametrize('B,T,Cin,N,KT,pad,Tout', CONV_CASES)
def test_gemm_conv(dev, B, T, Cin, N, KT, pad, Tout):
    from semi_tts_amd import ops
    x = rnd(B, T, Cin, seed=1)
    w = rnd(N, Cin, KT, scale=(Cin * KT) ** -0.5, seed=2)
    b = rnd(N, seed=3)
    ref = O.conv1d_cl(x.double(), w.double(), b.double(), pad)
    if Tout is not None:
        ref = ref[:, :Tout]
    wd = w.to(dev) if KT > 1 else w[:, :, 0].contiguous().to(dev)
    y = ops.gemm(x.to(dev), wd, pad=pad, Tout=Tout, bias=b.to(dev))
    err = maxdiff(y, ref)
    report('gemm_conv', B=B, T=T, Cin=Cin, N=N, KT=KT, err=err)
    assert y.shape == ref.shape and err < 2e-5


def test_gemm_epilogues(dev):
    from semi_tts_amd import ops
    B, T, Cin, N = 3, 17, 48, 72
    x, w, b = rnd(B, T, Cin, seed=1), rnd(N, Cin, 3, scale=0.1, seed=2), rnd(N, seed=3)
    mean, var = rnd(N, scale=0.2, seed=4), torch.rand(N) + 0.5
    g, beta = torch.rand(N) + 0.5, rnd(N, scale=0.1, seed=5)
    d = lambda t: t.to(dev).contiguous()
    conv = O.conv1d_cl(x, w, b, 1)
    # encoder order: conv -> BN -> ReLU
    y = ops.gemm(d(x), d(w), pad=1, bias=d(b), bn=(d(mean), d(var), d(g), d(beta)), bn_eps=1e-5, act_post='relu')
    ref = torch.relu((conv - mean) / torch.sqrt(var + 1e-5) * g + beta)
    assert maxdiff(y, ref) < 1e-5
    # CBHG order: conv -> ReLU -> BN, written into a column slice of a wider buffer
    buf = torch.zeros(B, T, 200, device=dev)
    ops.gemm(d(x), d(w), buf, pad=1, coff=100, act_pre='relu', bn=(d(mean), d(var), d(g), d(beta)), bn_eps=1e-3)
    ref = (torch.relu(conv - b) - mean) / torch.sqrt(var + 1e-3) * g + beta
    assert maxdiff(buf[:, :, 100:172], ref) < 1e-5
    assert float(buf[:, :, :100].abs().max()) == 0 and float(buf[:, :, 172:].abs().max()) == 0
    # max-pool(2, stride 1, pad 1)[:T] fused into the operand load
    prev = torch.cat([torch.full((B, 1, Cin), -float('inf')), x[:, :-1]], 1)
    ref = O.conv1d_cl(torch.maximum(prev, x), w, None, 1)
    assert maxdiff(ops.gemm(d(x), d(w), pad=1, pool_prev=True), ref) < 1e-5
    # highway: y = relu(Hx) * sigmoid(Tx) + x * (1 - sigmoid(Tx)); residual; dropout mask
    xs, wh, wt = rnd(40, N, seed=6), rnd(N, N, scale=0.1, seed=7), rnd(N, N, scale=0.1, seed=8)
    h = ops.gemm(d(xs), d(wh), bias=d(b), act_pre='relu')
    y = ops.gemm(d(xs), d(wt), bias=d(beta), act_pre='sigmoid', highway_h=h, res=d(xs))
    Hh, Tt = torch.relu(xs @ wh.t() + b), torch.sigmoid(xs @ wt.t() + beta)
    assert maxdiff(y, Hh * Tt + xs * (1 - Tt)) < 1e-5
    mask = (torch.rand(40, N) > 0.5).float() * 2
    y = ops.gemm(d(xs), d(wh), res=d(xs), mask=d(mask), act_pre='relu')
    assert maxdiff(y, (torch.relu(xs @ wh.t()) + xs) * mask) < 1e-5


def test_batchnorm_training_ops(dev):
    from semi_tts_amd import ops
    M, N = 300, 70
    x = rnd(M, 100, seed=1) * 2 + 0.5
    rm, rv = rnd(N, seed=2).to(dev), (torch.rand(N) + 0.5).to(dev)
    rm0, rv0 = rm.clone().cpu(), rv.clone().cpu()
    xd = x.to(dev)
    mean, var = ops.bn_stats(xd, 20, N, rm, rv, 0.99)
    xs = x[:, 20:90]
    assert maxdiff(mean, xs.mean(0)) < 1e-5 and maxdiff(var, xs.var(0, unbiased=False)) < 1e-4
    assert maxdiff(rm, 0.01 * rm0 + 0.99 * xs.mean(0)) < 1e-5
    assert maxdiff(rv, 0.01 * rv0 + 0.99 * xs.var(0, unbiased=True)) < 1e-4
    g, b = (torch.rand(N) + 0.5), rnd(N, seed=3)
    ops.bn_apply(xd, 20, N, mean, var, g.to(dev), b.to(dev), 1e-3, 'relu')
    ref = torch.relu((xs - xs.mean(0)) / torch.sqrt(xs.var(0, unbiased=False) + 1e-3) * g + b)
    assert maxdiff(xd[:, 20:90], ref) < 1e-4
    assert maxdiff(xd[:, :20], x[:, :20]) == 0 and maxdiff(xd[:, 90:], x[:, 90:]) == 0


@pytest.mark.parametrize('B,T,I,H', [(2, 5, 12, 8), (4, 12, 512, 256), (32, 43, 512, 256), (3, 7, 20, 12)])
def test_bilstm_sequence(dev, B, T, I, H):
    from semi_tts_amd import ops
    x = rnd(B, T, I, seed=1)
    W = {}
    for sfx in ('_l0', '_l0_reverse'):
        W['l.weight_ih' + sfx] = rnd(4 * H, I, scale=I ** -0.5, seed=2 + len(sfx))
        W['l.weight_hh' + sfx] = rnd(4 * H, H, scale=H ** -0.5, seed=3 + len(sfx))
        W['l.bias_ih' + sfx] = rnd(4 * H, scale=0.1, seed=4 + len(sfx))
        W['l.bias_hh' + sfx] = rnd(4 * H, scale=0.1, seed=5 + len(sfx))
    ref = torch.cat([O.lstm_layer(x, W, 'l', False), O.lstm_layer(x, W, 'l', True)], -1)
    out = torch.empty(B, T, 2 * H, device=dev)
    xd = x.to(dev)
    for rev, sfx in ((False, '_l0'), (True, '_l0_reverse')):
        xp = ops.gemm(xd, W['l.weight_ih' + sfx].to(dev), bias=W['l.bias_ih' + sfx].to(dev))
        ops.lstm_seq(xp, W['l.weight_hh' + sfx].to(dev), W['l.bias_hh' + sfx].to(dev), out, H if rev else 0, rev)
    err = maxdiff(out, ref)
    report('bilstm', B=B, T=T, H=H, err=err)
    assert err < 2e-5


@pytest.mark.parametrize('B,T,H', [(2, 6, 8), (4, 66, 80), (3, 258, 80), (2, 9, 130)])
def test_bigru_sequence(dev, B, T, H):
    from semi_tts_amd import ops
    x = rnd(B, T, H, seed=1)
    W = {}
    for sfx in ('_l0', '_l0_reverse'):
        W['g.weight_ih' + sfx] = rnd(3 * H, H, scale=H ** -0.5, seed=2 + len(sfx))
        W['g.weight_hh' + sfx] = rnd(3 * H, H, scale=H ** -0.5, seed=3 + len(sfx))
        W['g.bias_ih' + sfx] = rnd(3 * H, scale=0.1, seed=4 + len(sfx))
        W['g.bias_hh' + sfx] = rnd(3 * H, scale=0.1, seed=5 + len(sfx))
    ref = torch.cat([O.gru_layer(x, W, 'g', False), O.gru_layer(x, W, 'g', True)], -1)
    d = lambda k: W[k].to(dev)
    xd = x.to(dev)
    gi_f = ops.gemm(xd, d('g.weight_ih_l0'), bias=d('g.bias_ih_l0'))
    gi_b = ops.gemm(xd, d('g.weight_ih_l0_reverse'), bias=d('g.bias_ih_l0_reverse'))
    out = torch.empty(B, T, 2 * H, device=dev)
    ops.gru_seq(gi_f, gi_b, d('g.weight_hh_l0'), d('g.weight_hh_l0_reverse'), d('g.bias_hh_l0'),
                d('g.bias_hh_l0_reverse'), out)
    err = maxdiff(out, ref)
    report('bigru', B=B, T=T, H=H, err=err)
    assert err < 2e-5


# ------------------------------------------------------------------------------------ VQ
@pytest.mark.parametrize('name', ['vq_l2_native', 'vq_l2_512', 'vq_l2_temp'])
def test_vq_l2_against_reference(dev, name):
    from semi_tts_amd.embed import L2Embedding
    W, A, meta = load_golden(name)
    V = meta['V']
    use_attr = 'proj_attr.weight' in W
    cb = L2Embedding(V, False, 'normal', 64, 0, 0, float(W['temp'][0]), 0, True)
    if use_attr:      # attach the attribute table from the fixture (data), not from the reference tree
        cb.use_phn_attr = True
        cb.phn_attr = torch.nn.Embedding.from_pretrained(W['phn_attr.weight'], freeze=True, padding_idx=0)
        cb.proj_attr = torch.nn.Linear(31, 16)
        cb.learnable_table = torch.nn.Parameter(torch.zeros(V, 48))
    cb.load_state_dict(W)
    cb = cb.to(dev).eval()
    p, out, _, _ = cb(A['x'].to(dev))
    idx = cb.last_idx.cpu()
    n_bad = int((idx != A['idx']).sum())
    report('vq_l2', name=name, mismatches=n_bad, err_p=maxdiff(p, A['p_code']))
    assert torch.equal(idx, A['idx']), 'VQ code indices must be bit-exact (%d differ)' % n_bad
    # sims are O(100) (squared distances of 64-dim vectors), so one fp32 ulp of a sim (7.6e-6)
    # moves a softmax entry by ~1e-5 relative: 5e-5 absolute on p in [0,1]
    assert maxdiff(p, A['p_code']) < 5e-5
    assert maxdiff(out, A['new_latent']) < 1e-6
    if 'inference' in A:
        assert maxdiff(cb.inference(A['txt'].to(dev)), A['inference']) < 1e-6
        assert maxdiff(cb.embedding.weight, A['table']) < 1e-6


def test_vq_seperate_against_reference(dev):
    from semi_tts_amd.embed import SeperateEmbedding
    W, A, meta = load_golden('vq_seperate')
    cb = SeperateEmbedding(43, False, 'normal', 64, 0, 0, 1, 0, True)
    cb.use_phn_attr = True
    cb.phn_attr = torch.nn.Embedding.from_pretrained(W['phn_attr.weight'], freeze=True, padding_idx=0)
    cb.proj_attr = torch.nn.Linear(31, 16)
    cb.embedding = torch.nn.Embedding(43, 48)
    cb.load_state_dict(W)
    cb = cb.to(dev).eval()
    p, out, _, _ = cb(A['x'].to(dev))
    assert torch.equal(cb.last_idx.cpu(), A['idx'])
    assert maxdiff(p, A['p_code']) < 5e-6 and maxdiff(out, A['new_latent']) < 1e-6
    assert maxdiff(cb.inference(A['txt'].to(dev)), A['inference']) < 1e-6


@pytest.mark.parametrize('V', [43, 512])
def test_vq_full_size_properties(dev, V):
    """C3 size (32 x 129 vectors): indices against the oracle, plus size-independent properties:
    exact codes map to themselves, quantising twice is idempotent, p rows sum to 1."""
    from semi_tts_amd import ops
    g = torch.Generator().manual_seed(3)
    table = torch.randn(V, 64, generator=g)
    x = torch.randn(32, 129, 64, generator=g)
    x[0, :V if V < 129 else 129] = table[:129 if V >= 129 else V]
    temp = torch.tensor([1.0])
    p_ref, idx_ref, out_ref, _ = VQ.l2_forward({'learnable_table': table, 'temp': temp}, x)
    p, idx, out = ops.vq_l2(x.to(dev), table.to(dev), temp.to(dev))
    n_bad = int((idx.cpu() != idx_ref).sum())
    report('vq_full', V=V, mismatches=n_bad)
    assert n_bad == 0
    assert maxdiff(p, p_ref) < 5e-5 and maxdiff(out, out_ref) < 1e-6
    k = min(V, 129)
    assert torch.equal(idx[0, :k].cpu(), torch.arange(k))
    _, idx2, _ = ops.vq_l2(ops.gather_rows(table.to(dev), idx), table.to(dev), temp.to(dev))
    assert torch.equal(idx2, idx)
    assert float((p.sum(-1) - 1).abs().max()) < 1e-5


# ------------------------------------------------------------------------------------ module level
TTS_CASES = ['tts_tiny_infer', 'tts_tiny_infer_nodrop', 'tts_tiny_train_tf', 'tts_tiny_eval_tf',
             'tts_tiny_sched', 'tts_tiny_partial', 'tts_tiny_quirk',
             # decoder / encoder variants no shipped YAML reaches: speaker-conditioned memory, pre-training, 2-layer encoder LSTM
             'tts_tiny_concat', 'tts_tiny_add', 'tts_tiny_pretrain', 'tts_tiny_enc2', 'tts_tiny_dropin', 'tts_tiny_noloc',
             'tts_tiny_nosum', 'tts_tiny_encdrop',
             # normalised prenet (prenet_norm_type LayerNorm / BatchNorm1d): eval, teacher-forced training, scheduled sampling
             'tts_tiny_preln_infer', 'tts_tiny_preln_train', 'tts_tiny_prebn_infer', 'tts_tiny_prebn_train', 'tts_tiny_prebn_sched',
             'tts_tiny_preln_sched', 'tts_tiny_prebn_partial']


@pytest.mark.parametrize('name', TTS_CASES)
def test_tacotron2_against_reference_golden(dev, name):
    """HIP Tacotron2.forward vs outputs recorded from the real reference (tiny dimensions),
    replaying the reference's own dropout masks and teacher-forcing coin flips."""
    from semi_tts_amd.module import plan_decode
    W, A, meta = load_golden(name)
    hp = meta['hp']
    m = tiny_tacotron(meta, W, dev)
    m.train(meta['training'])
    txt, spk = A['txt_embed'].to(dev), A['spkr_embed'].to(dev)
    B = txt.shape[0]
    is_int = meta['teacher'] is not None
    teacher = meta['teacher'] if is_int else A['teacher'].to(dev)
    Bt = B if is_int else A['teacher'].shape[0]
    steps, src = plan_decode(is_int, teacher if is_int else teacher.shape[1], Bt, B, hp['n_frames_per_step'],
                             meta['tf_rate'], hp['drop_dec_in'], meta['unpair_max_frame'], coin_source(A['coins']))
    masks = masks_to(split_masks(A.get('mask', []), hp, meta['training'], meta['tf_rate'], B, Bt, steps, src,
                                 hp['prenet_dim']), dev)
    import semi_tts_amd.module as M
    saved = np.random.rand
    np.random.rand = coin_source(A['coins'])          # the host RNG the reference consumes
    try:
        with torch.no_grad():
            mel, lin, align, stop = m(txt, None, teacher, spk, tf_rate=meta['tf_rate'],
                                      unpair_max_frame=meta['unpair_max_frame'], _masks=masks)
    finally:
        np.random.rand = saved
    errs = dict(mel=maxdiff(mel, A['mel']), lin=maxdiff(lin, A['linear']), align=maxdiff(align, A['align']),
                stop=maxdiff(stop, A['stop']))
    report('tts_golden', name=name, **errs)
    assert mel.shape == A['mel'].shape and lin.shape == A['linear'].shape
    assert errs['mel'] < 1e-4 and errs['align'] < 1e-5 and errs['stop'] < 1e-4 and errs['lin'] < 2e-4
    if meta['training']:
        import json
        keys = json.loads(bytes(A['post_keys']).decode())
        sd = m.state_dict()
        for k, v in zip(keys, A['post']):
            assert maxdiff(sd[k], v) < 1e-4, k


def test_full_size_c1_against_reference(dev):
    """Full dimensions, B=4, T=66, L=12 (BASELINE config 1) against the real reference's output
    for the same seeded synthetic weights: free-running inference and teacher forcing."""
    _, A, meta = load_golden('tts_full_c1')
    m = full_tacotron(dev, seed=meta['seed'])
    txt, spk = A['txt_embed'].to(dev), A['spkr_embed'].to(dev)
    with torch.no_grad():
        mel, lin, align, stop = m(txt, None, meta['T'], spk, tf_rate=0.0)
        mel_t, lin_t, align_t, stop_t = m(txt, None, A['teacher'].to(dev), spk, tf_rate=1.0)
    errs = dict(mel_infer=maxdiff(mel, A['mel_infer']), align_infer=maxdiff(align, A['align_infer']),
                stop_infer=maxdiff(stop, A['stop_infer']), lin_infer=maxdiff(lin[:, ::7, ::41], A['linear_infer_s']),
                mel_tf=maxdiff(mel_t, A['mel_tf']), align_tf=maxdiff(align_t, A['align_tf']),
                lin_tf=maxdiff(lin_t[:, ::7, ::41], A['linear_tf_s']))
    report('tts_full_c1', **errs)
    # north-star bound: 1e-3 max-abs on mel; measured error is summation-order noise
    assert errs['mel_infer'] < 1e-3 and errs['mel_tf'] < 1e-3
    assert errs['mel_infer'] < 5e-5 and errs['mel_tf'] < 5e-5
    assert errs['align_infer'] < 1e-5 and errs['align_tf'] < 1e-5
    assert errs['lin_infer'] < 2e-4 and errs['lin_tf'] < 2e-4


def _oracle_weights(m):
    return {k: v.detach().cpu() for k, v in m.state_dict().items()}


@pytest.mark.parametrize('B,L,T', [(1, 1, 3), (2, 2, 9), (3, 200, 12), (17, 43, 30), (33, 7, 6), (5, 400, 6)])
def test_edge_shapes_against_oracle(dev, B, L, T):
    """smallest, ragged and long inputs through the whole HIP Tacotron2.forward (full dims): one position, one utterance,
    batches straddling the 16-row tiles, text far longer than the headline shape"""
    m = full_tacotron(dev, seed=5)
    g = torch.Generator().manual_seed(B * 1000 + L)
    txt, spk = torch.randn(B, L, 64, generator=g), torch.randn(B, 128, generator=g)
    with torch.no_grad():
        mel, lin, align, stop = m(txt.to(dev), None, T, spk.to(dev), tf_rate=0.0)
        mel_r, lin_r, align_r, _ = O.tacotron2_forward(_oracle_weights(m), txt, T, spk, full_hp(0.0))
    errs = dict(mel=maxdiff(mel, mel_r), lin=maxdiff(lin, lin_r), align=maxdiff(align, align_r))
    report('tts_edge', B=B, L=L, T=T, **errs)
    assert errs['mel'] < 2e-5 and errs['lin'] < 2e-5 and errs['align'] < 1e-5


def test_long_text_runs_split_and_the_unsplit_step_refuses_it(dev):
    """Text far beyond the headline length (L = 900): the decode loop's split attention holds only a position range (pre part)
    or the energies (fin part) in LDS, so it runs and matches the oracle; the one-launch step keeps an utterance's conv features
    in LDS and refuses the same text with an error from the C ABI -- not a wrong answer."""
    from semi_tts_amd import ops
    m = full_tacotron(dev, seed=5)
    g = torch.Generator().manual_seed(900)
    txt, spk = torch.randn(2, 900, 64, generator=g), torch.randn(2, 128, generator=g)
    with torch.no_grad():
        mel, lin, align, _ = m(txt.to(dev), None, 6, spk.to(dev), tf_rate=0.0)
        mel_r, lin_r, align_r, _ = O.tacotron2_forward(_oracle_weights(m), txt, 6, spk, full_hp(0.0))
    errs = dict(mel=maxdiff(mel, mel_r), lin=maxdiff(lin, lin_r), align=maxdiff(align, align_r))
    report('tts_long_text', L=900, **errs)
    assert errs['mel'] < 2e-5 and errs['lin'] < 2e-5 and errs['align'] < 1e-5
    B, L, A, E = 2, 900, 256, 512
    z = lambda *sh: torch.zeros(*sh, device=dev)
    with pytest.raises(RuntimeError, match='LDS'):
        ops.attn_step(z(B, A), z(B, L, A), z(B, L, E), z(B, L), z(B, L), z(B, L), z(B, L), z(32, 2, 31), z(A, 32), z(A), z(B, E))


def test_headline_shape_c2_against_oracle(dev):
    """BASELINE config 2 shape (B=32, T=258 -> 86 steps, L=43): whole Tacotron2.forward,
    inference, against the CPU oracle on the same inputs; mel within 1e-3 (north star)."""
    from semi_tts_amd.synthetic import synthetic_batch
    m = full_tacotron(dev, seed=99)
    txt, spk, _ = synthetic_batch(32, 43, 258)
    txt, spk = torch.from_numpy(txt), torch.from_numpy(spk)
    with torch.no_grad():
        mel, lin, align, stop = m(txt.to(dev), None, 258, spk.to(dev), tf_rate=0.0)
        # the unsplit attention launch and other part counts must give the same values up to re-association
        for split, fin in ((False, 2), (True, 1), (True, 4)):
            m.decoder.attn_split, m.decoder.attn_fin_parts = split, fin
            mel_o, _, align_o, _ = m(txt.to(dev), None, 258, spk.to(dev), tf_rate=0.0)
            report('tts_c2_attn_variants', split=int(split), fin=fin, mel=maxdiff(mel_o, mel), align=maxdiff(align_o, align))
            assert maxdiff(mel_o, mel) < 2e-5 and maxdiff(align_o, align) < 1e-5
        m.decoder.attn_split, m.decoder.attn_fin_parts = True, 2
        # the query projection handed to the attention fin part INSIDE one launch (8-byte {value, tag} granules) is the same
        # arithmetic as the two launches: bit-identical outputs, and again on a second pass (the tags of the first must not leak)
        m.decoder.attn_pq_in_fin = False
        mel_2, _, align_2, _ = m(txt.to(dev), None, 258, spk.to(dev), tf_rate=0.0)
        m.decoder.attn_pq_in_fin = True
        mel_1, _, align_1, _ = m(txt.to(dev), None, 258, spk.to(dev), tf_rate=0.0)
        assert torch.equal(mel_2, mel) and torch.equal(align_2, align) and torch.equal(mel_1, mel) and torch.equal(align_1, align)
    torch.set_num_threads(8)
    with torch.no_grad():
        mel_r, lin_r, align_r, stop_r = O.tacotron2_forward(_oracle_weights(m), txt, 258, spk, full_hp(0.0))
    errs = dict(mel=maxdiff(mel, mel_r), lin=maxdiff(lin, lin_r), align=maxdiff(align, align_r), stop=maxdiff(stop, stop_r))
    report('tts_c2', **errs)
    assert mel.shape == (32, 258, 80) and lin.shape == (32, 258, 1025) and align.shape == (32, 86, 43)
    assert errs['mel'] < 1e-3 and errs['lin'] < 1e-3 and errs['align'] < 1e-4
    # size-independent properties: alignments are distributions, stop is constant within a step
    assert float((align.sum(-1) - 1).abs().max()) < 1e-5
    st = stop.view(32, 86, 3)
    assert float((st - st[:, :, :1]).abs().max()) == 0.0


def test_decode_is_deterministic_and_graph_replay_matches(dev):
    """Two eager runs are bit-identical; a hipGraph capture of the decode loop replays to the
    same bits (no hidden host state in the loop)."""
    from semi_tts_amd import ops
    m = full_tacotron(dev, seed=7)
    g = torch.Generator().manual_seed(1)
    B, L, T = 4, 9, 30
    mem = torch.randn(B, L, 512, generator=g).to(dev)
    spk = torch.randn(B, 128, generator=g).to(dev)
    with torch.no_grad():
        a = m.decoder(mem, None, T, spk)[0].clone()
        b = m.decoder(mem, None, T, spk)[0].clone()
    assert torch.equal(a, b)
    from semi_tts_amd.runtime import GraphedDecoder
    gd = GraphedDecoder(m.decoder, B, L, T, dev).capture()
    out = gd(mem, spk)[0]
    torch.cuda.synchronize()
    assert torch.equal(out, a)
    out.zero_()
    out2 = gd(mem, spk)[0]                               # replay overwrites the same static buffers
    torch.cuda.synchronize()
    assert out2.data_ptr() == out.data_ptr() and torch.equal(out2, a)
    # the graph replays with the weights packed at capture time; after a weight change refresh_weights() re-packs
    bias0 = m.decoder.proj.linear.bias.detach().clone()
    with torch.no_grad():
        m.decoder.proj.linear.bias.add_(0.25)
    gd.refresh_weights()
    out3 = gd(mem, spk)[0].clone()
    torch.cuda.synchronize()
    with torch.no_grad():
        c = m.decoder(mem, None, T, spk)[0].clone()           # eager, served from the packed-weight cache
        assert torch.equal(out3, c) and not torch.equal(out3, a)
        m.decoder.proj.linear.bias.copy_(bias0)               # an in-place weight change invalidates the cache by itself
        d = m.decoder(mem, None, T, spk)[0]
    assert torch.equal(d, a)


def test_whole_forward_graph_replay_matches_eager(dev):
    """GraphedTacotron2: encoder + decode loop + CBHG postnet + linear captured into one hipGraph; replay is bit-identical to
    the eager forward with the same masks."""
    from semi_tts_amd.runtime import GraphedTacotron2
    m = full_tacotron(dev, seed=7, prenet_dropout=0.5)
    g = torch.Generator().manual_seed(2)
    B, L, T = 5, 11, 24
    txt, spk = torch.randn(B, L, 64, generator=g).to(dev), torch.randn(B, 128, generator=g).to(dev)
    gt = GraphedTacotron2(m, B, L, T, dev).capture()
    mel, lin, align, stop = gt(txt, spk, redraw=True)
    torch.cuda.synchronize()
    with torch.no_grad():
        mel_e, lin_e, align_e, stop_e = m(txt, None, T, spk, tf_rate=0.0, _masks={'own': gt.own_mask})
    assert torch.equal(mel, mel_e) and torch.equal(lin, lin_e) and torch.equal(align, align_e) and torch.equal(stop, stop_e)
    a = lin.clone()
    # the intermediates the forward allocated during the capture (encoder / postnet activations, GEMM workspaces) were freed
    # when it returned, and the graph still writes to them: they live in the graph's private memory pool, so other tensors of
    # the process never land on them -- allocate and poison plenty of memory, replay, and look at both sides
    junk = [torch.full((n,), float('nan'), device=dev) for n in (1 << 12, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 3 << 20, 5 << 18) * 3]
    lin2 = gt(redraw=False)[1]
    torch.cuda.synchronize()
    assert torch.equal(a, lin2)
    assert all(bool(torch.isnan(j).all()) for j in junk)


def test_cpu_tensor_is_refused():
    """the product path has no CPU fallback: CPU tensors raise"""
    from semi_tts_amd import ops
    with pytest.raises(RuntimeError):
        ops.linear_small(torch.zeros(2, 8), torch.zeros(4, 8))


# ------------------------------------------------------------------------------------ callers (H2) and long form (C5)
def _full_vqvae(dev, bone='l2', seed=77):
    import yaml
    from semi_tts_amd.synthetic import load_synthetic
    from semi_tts_amd.vqvae import VQVAE
    import os
    from conftest import REPO
    cfg = yaml.safe_load(open(os.path.join(REPO, 'config', 'semi-single-spkr-paired-data.yaml')))['model']
    cfg['codebook'].update(bone=bone, phn_attr_pth='', proj_attr=None)   # the attribute csv lives in the reference tree
    cfg['decoder']['decoder']['prenet_dropout'] = 0.0
    m = VQVAE(80, 1025, 43, 109, **cfg)
    load_synthetic(m, seed)
    return m.to(dev).eval()


def test_vqvae_text_to_speech_against_oracle(dev):
    """VQVAE.text_to_speech (codebook.inference -> speaker embedding -> Tacotron2) with the reference's
    argument list and 8-tuple, paired-only and text-to-text (unpaired text) batches"""
    m = _full_vqvae(dev)
    W = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    Wt = {k[4:]: v for k, v in W.items() if k.startswith('tts.')}
    Wc = {k[9:]: v for k, v in W.items() if k.startswith('codebook.')}
    g = torch.Generator().manual_seed(3)
    text = torch.randint(3, 43, (3, 9), generator=g)
    sid = torch.randint(0, 109, (3,), generator=g)
    frames = 30
    with torch.no_grad():
        out = m.text_to_speech(text.to(dev), sid.to(dev), None, None, None, None, frames, None, tf_rate=0.0)
    assert len(out) == 8 and all(o is None for o in out[4:])
    lat = VQ.l2_inference(Wc, text)
    spk = W['spkr_embed.weight'][sid]
    mel_r, lin_r, al_r, st_r = O.tacotron2_forward(Wt, lat, frames, spk, full_hp(0.0))
    errs = dict(mel=maxdiff(out[0], mel_r), lin=maxdiff(out[1], lin_r), align=maxdiff(out[2], al_r))
    report('vqvae_tts', **errs)
    assert errs['mel'] < 5e-5 and errs['lin'] < 2e-4 and errs['align'] < 1e-5


def test_specgram_generator_writes_reference_style_files(dev, tmp_path):
    """H2: the gen_specgram counterpart decodes mel_len + 40 frames and writes -mel/-spec/-align .npy"""
    import types
    import yaml
    import os
    from conftest import REPO
    from semi_tts_amd.solver import SpecgramGenerator, INFERENCE_MARGIN_FRAMES
    config = yaml.safe_load(open(os.path.join(REPO, 'config', 'supervised.yaml')))
    paras = types.SimpleNamespace(name='t', logdir=str(tmp_path), load=None, seed=1, cpu=False, verbose=False,
                                  batch_size=2, frames=24, n_batches=1)
    s = SpecgramGenerator(config, paras, 'test')
    s.load_data()
    s.set_model()
    n = s.exec()
    assert n == 2
    out = os.path.join(str(tmp_path), 't_0k')
    mel = np.load(os.path.join(out, 'utt00000-mel.npy'))
    spec = np.load(os.path.join(out, 'utt00000-spec.npy'))
    ali = np.load(os.path.join(out, 'utt00001-align.npy'))
    T = 24 + 3 + INFERENCE_MARGIN_FRAMES             # 24 is a multiple of r=3: a full extra group is padded
    T -= T % 3
    assert mel.shape == (T, 80) and spec.shape == (T, 1025) and mel.dtype == np.float32
    assert ali.shape == (int(3 * 6.0) // 3, 3)        # text length 4 with the last token 0 -> 3 real tokens
    assert np.isfinite(mel).all() and np.isfinite(spec).all()


def test_long_form_c5_shape(dev):
    """BASELINE config 5 shape: B=64, L=171, 1026+40 frames -> 355 decode steps.  The decode is causal,
    so its first 8 steps must equal an 8-step decode by the oracle; plus size-independent properties."""
    from semi_tts_amd.synthetic import synthetic_batch
    m = full_tacotron(dev, seed=5)
    B, L, T = 64, 171, 1066
    txt, spk, _ = synthetic_batch(B, L, 3, seed=9)
    txt, spk = torch.from_numpy(txt), torch.from_numpy(spk)
    with torch.no_grad():
        mem = m.encoder(txt.to(dev), None)
        mel, align, stop = m.decoder(mem, None, T, spk.to(dev), tf_rate=0.0)
    assert mel.shape == (B, 1065, 80) and align.shape == (B, 355, L)
    assert bool(torch.isfinite(mel).all())
    assert float((align.sum(-1) - 1).abs().max()) < 1e-5
    W = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    torch.set_num_threads(8)
    with torch.no_grad():
        mel_r, al_r, _ = O.decoder_forward(W, mem.cpu(), 24, spk, full_hp(0.0))
    errs = dict(mel=maxdiff(mel[:, :24], mel_r), align=maxdiff(align[:, :8], al_r))
    report('c5_prefix', **errs)
    assert errs['mel'] < 5e-5 and errs['align'] < 1e-5


# ------------------------------------------------------------------------------------ next row (8f-1): mean_forward
def test_vq_mean_forward_against_reference_golden(dev):
    """Run-length merge + blank filter of VQ codes on device vs the reference's recorded outputs (incl. the
    all-blank utterance -> None case and runs longer than max_frames_per_phn)."""
    from semi_tts_amd import autograd as AG
    _, A, meta = load_golden('vq_mean_forward')
    for ci in range(meta['n_cases']):
        idx, lat = A['idx%d' % ci], A['lat%d' % ci]
        p = torch.nn.functional.one_hot(idx, 43).float()
        out = AG.mean_forward(p.to(dev), lat.to(dev), meta['max_frames_per_phn'])
        if ('none%d' % ci) in A:
            assert out is None
            continue
        got, ln = out
        assert torch.equal(ln.cpu(), A['len%d' % ci])
        assert got.shape == A['out%d' % ci].shape
        assert maxdiff(got, A['out%d' % ci]) < 1e-6


def test_vq_mean_forward_c3_size_and_gradient(dev):
    # C3 shape: B=32 utterances of 129 ASR frames, V=43, D=64; few distinct codes so that runs, caps and blanks all occur
    from semi_tts_amd import autograd as AG
    g = torch.Generator().manual_seed(3)
    B, T, D, V, maxf = 32, 129, 64, 43, 3
    idx = torch.randint(0, 5, (B, T), generator=g)
    idx = torch.where(torch.rand(B, T, generator=g) < 0.6, idx.roll(1, 1), idx)      # longer runs
    idx[:, 0] = torch.randint(1, 5, (B,), generator=g)                                # no all-blank utterance
    p = torch.rand(B, T, V, generator=g) * 0.1
    p.scatter_(2, idx.unsqueeze(-1), 1.0)
    lat = torch.randn(B, T, D, generator=g)
    ref = VQ.mean_forward(idx.numpy(), lat.numpy(), maxf)
    lat_d = lat.to(dev).requires_grad_()
    got, ln = AG.mean_forward(p.to(dev), lat_d, maxf)
    assert np.array_equal(ln.cpu().numpy(), ref[1])
    assert maxdiff(got, torch.from_numpy(ref[0])) < 1e-6
    # properties: every kept segment has 1..maxf+1 frames, lengths <= T, padding rows are exactly zero
    for b in range(B):
        assert float(got[b, int(ln[b]):].abs().max()) == 0.0 if int(ln[b]) < got.shape[1] else True
    # gradient: d/dlatent of sum(out * w) = w[seg(t)] / len(seg(t))
    w = torch.randn(*got.shape, generator=g)
    (got * w.to(dev)).sum().backward()
    lat_r = lat.double().requires_grad_()
    tot = 0
    for b in range(B):       # float64 restatement of the segmentation for the gradient reference
        row = idx[b].tolist()
        last_idx, last_pos, n = row[0], 0, 0
        for t, i in enumerate(row):
            if last_idx != i or (t - last_pos) > maxf:
                if last_idx != 0:
                    tot = tot + (lat_r[b, last_pos:t].mean(0) * w[b, n].double()).sum()
                    n += 1
                last_idx, last_pos = i, t
        if last_idx != 0:
            tot = tot + (lat_r[b, last_pos:].mean(0) * w[b, n].double()).sum()
    tot.backward()
    assert maxdiff(lat_d.grad, lat_r.grad) < 1e-6


# ------------------------------------------------------------------------------------ next row (8f-2): speech encoder
@pytest.mark.parametrize('name', ['asr_tiny_eval', 'asr_tiny_train', 'asr_tiny_ln_eval', 'asr_tiny_ln_train', 'asr_tiny_uni_eval',
                                  'asr_tiny_uni_train'])
def test_ctc_encoder_against_reference_golden(dev, name):
    import json
    from semi_tts_amd.asr import CTC
    W, A, meta = load_golden(name)
    m = CTC(meta['in_dim'], meta['out_dim'], **meta['cfg'])
    m.load_state_dict(W)
    m = m.to(dev).train(meta['training'])
    with torch.no_grad():
        y = m(A['x'].to(dev))
    err = maxdiff(y, A['y'])
    report('asr_golden', name=name, err=err)
    assert y.shape == A['y'].shape and err < 2e-5
    if meta['training']:
        sd = m.state_dict()
        for k, v in zip(json.loads(bytes(A['post_keys']).decode()), A['post']):
            assert maxdiff(sd[k], v) < 1e-5, k


def test_speech_to_text_c3_shape_against_oracle(dev):
    """Full-size CTC encoder (6 conv layers of 512 incl. the stride-2 one, 2-layer BiLSTM(256), Linear -> 64), the L2
    codebook and the run-length merge, B=8 utterances of 258 frames, vs the oracle composition on CPU."""
    import yaml, os
    from oracle import asr_oracle as AO
    from semi_tts_amd.synthetic import load_synthetic
    from semi_tts_amd.vqvae import VQVAE
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, 'config', 'semi-single-spkr-paired-data.yaml')))['model']
    cfg['codebook'].update(phn_attr_pth='', proj_attr=None)
    m = VQVAE(80, 1025, 43, 109, **cfg)
    load_synthetic(m, 77)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(5)
    mel, umel = torch.rand(5, 258, 80, generator=g), torch.rand(3, 200, 80, generator=g)
    with torch.no_grad():
        pp, pl, up, ul, ulen, _, _ = m.speech_to_text(mel.to(dev), umel.to(dev))
    Wd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    allmel = torch.zeros(8, 258, 80)
    allmel[:5], allmel[5:, :200] = mel, umel
    enc = AO.ctc_forward({k[4:]: v for k, v in Wd.items() if k.startswith('asr.')}, allmel, cfg['encoder'])
    cbW = {k[9:]: v for k, v in Wd.items() if k.startswith('codebook.')}
    p_ref, idx_ref, out_ref, table = VQ.l2_forward(cbW, enc)
    assert pp.shape == (5, 129, 43) and pl.shape == (5, 129, 64)
    # the codes are an argmax over distances computed from a ~1e-5-accurate encoder output: compare where decisive
    p_all = torch.cat([pp, up], 0).cpu()
    idx_hip = p_all.argmax(-1)
    top2 = p_ref.topk(2, -1).values
    decisive = (top2[..., 0] - top2[..., 1]) > 2e-5          # p is near-uniform with synthetic weights (entries ~1/43)
    assert float(decisive.float().mean()) > 0.8
    assert bool((idx_hip[decisive] == idx_ref[decisive]).all())
    errs = dict(p=maxdiff(p_all[decisive], p_ref[decisive]))
    report('speech_to_text_c3', decisive=float(decisive.float().mean()), **errs)
    assert errs['p'] < 1e-5
    # run-length merge of the unpaired part: oracle segmentation on the HIP codes and HIP latents
    mf = VQ.mean_forward(idx_hip[5:].numpy(), table[idx_hip[5:]].numpy(), cfg['max_frames_per_phn'])
    if mf is None:
        assert ul is None
    else:
        assert np.array_equal(ulen.cpu().numpy(), mf[1]) and maxdiff(ul, torch.from_numpy(mf[0])) < 1e-5


# ------------------------------------------------------------------------------------ round 3: packed-table cache, graphed speech_to_text
@pytest.mark.parametrize('V', [43, 512])
def test_vq_packed_table_search_is_bit_identical_and_cache_follows_the_weights(dev, V):
    """st_vq_pack_table + st_vq_l2_packed_fwd (table packed once per version) == st_vq_l2_fwd (packed per call) bit for bit;
    L2Embedding's cache is rebuilt when the table's parameter changes in place (an optimiser step, load_state_dict)."""
    from semi_tts_amd import ops
    from semi_tts_amd.embed import L2Embedding
    g = torch.Generator().manual_seed(11)
    table = torch.randn(V, 64, generator=g).to(dev)
    x = torch.randn(7, 129, 64, generator=g).to(dev)
    temp = torch.tensor([1.0], device=dev)
    p0, i0, o0 = ops.vq_l2(x, table, temp)
    p1, i1, o1 = ops.vq_l2(x, table, temp, packed=ops.vq_pack_table(table))
    assert torch.equal(i0, i1) and torch.equal(p0, p1) and torch.equal(o0, o1)
    cb = L2Embedding(V, False, 'normal', 64, 0, 0, 1.0, 0, True).to(dev).eval()
    with torch.no_grad():
        cb.learnable_table.copy_(table)
        pa, oa, _, _ = cb(x)
        ia = cb.last_idx.clone()
        ent = cb.__dict__['_table_cache']
        cb(x)
        assert cb.__dict__['_table_cache'] is ent                       # same weights: the packed table is reused
        assert torch.equal(ia, i0) and torch.equal(pa, p0)
        cb.learnable_table.mul_(-1.0)                                   # in-place update -> new version -> new table
        pb, ob, _, _ = cb(x)
        assert cb.__dict__['_table_cache'] is not ent
        p2, i2, o2 = ops.vq_l2(x, -table, temp)
        assert torch.equal(cb.last_idx, i2) and torch.equal(pb, p2) and torch.equal(ob, o2)


def test_graphed_speech_to_text_equals_eager(dev):
    """runtime.GraphedSpeechToText (speech encoder -> codebook -> run-length merge as ONE hipGraph) returns what
    VQVAE.speech_to_text returns, bit for bit, also on a replay with new inputs"""
    import yaml, os
    from semi_tts_amd.runtime import GraphedSpeechToText
    from semi_tts_amd.synthetic import load_synthetic
    from semi_tts_amd.vqvae import VQVAE
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, 'config', 'semi-single-spkr-paired-data.yaml')))['model']
    cfg['codebook'].update(phn_attr_pth='', proj_attr=None)
    m = VQVAE(80, 1025, 43, 109, **cfg)
    load_synthetic(m, 77)
    m = m.to(dev).eval()
    gs = GraphedSpeechToText(m, 5, 258, dev, B_unpair=3).capture()
    for seed in (5, 6):
        g = torch.Generator().manual_seed(seed)
        mel, umel = torch.rand(5, 258, 80, generator=g).to(dev), torch.rand(3, 200, 80, generator=g).to(dev)
        with torch.no_grad():
            ref = m.speech_to_text(mel, umel)
        got = gs(mel, umel)
        for name, a, b in zip(('pair_prob', 'pair_latent', 'unpair_prob', 'unpair_latent', 'unpair_len'), got, ref):
            assert (a is None) == (b is None), name
            if a is not None:
                assert torch.equal(a, b), name


@pytest.mark.parametrize('B,L,Q,A,E,parts', [(64, 171, 1024, 256, 512, 4), (3, 171, 1024, 256, 512, 4), (5, 130, 64, 32, 64, 2),
                                             (2, 300, 128, 64, 128, 8), (7, 17, 48, 16, 128, 4)])
def test_query_projection_and_fin_part_over_position_ranges_in_one_launch(dev, B, L, Q, A, E, parts):
    """st_query_attn_rng_fwd (long texts, BASELINE config 5): query projection + fin part over position ranges + combine in ONE launch
    -- pq and the ranges' (max, sum, partial context) travel as {value, tag} granules inside the launch -- against the packed
    linear + st_attn_fin_split_fwd (same split, combine as a second launch) and against float64; consecutive epochs on the same
    granule buffers; the status word stays clear."""
    from semi_tts_amd import _lib, ops
    lib = _lib.load()
    assert lib.st_query_attn_rng_fits(B, A, parts) == 1
    wq, hq = rnd(A, Q, scale=Q ** -0.5, seed=1), rnd(B, Q, seed=2)
    S, mem = rnd(B, L, A, seed=3), rnd(B, L, E, seed=4)
    w_cum, v = torch.rand(B, L, generator=torch.Generator().manual_seed(5)), rnd(1, A, seed=6)
    d = [t.to(dev) for t in (wq, hq, S, mem, w_cum, v)]
    packed = ops.pack_weight([d[0]], [Q], A)
    h_t = ops.tile_rows(d[1])
    pq = torch.empty(B, A, device=dev)
    ops.skinny_linear_packed(packed, ops.t16_view(h_t, K=Q), 16 * ops.kb16(Q), B, A, y=pq)
    w1, c1, x1 = (torch.empty(B, L, device=dev), torch.empty(B, L, device=dev), torch.empty(B, E, device=dev))
    ops.attn_fin_split(pq, d[2], d[3], d[4], d[5], w1, c1, x1, parts)
    e = torch.tanh((hq.double() @ wq.double().t()).unsqueeze(1) + S.double()).matmul(v.double().view(-1))
    w_r = torch.softmax(e, -1)
    x_r = torch.bmm(w_r.unsqueeze(1), mem.double()).squeeze(1)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    gran = xchg = None
    for epoch in (1, 2, 3):
        w2, c2 = torch.full_like(w1, float('nan')), torch.full_like(c1, float('nan'))
        x_t = torch.zeros(ops.t16_floats(B, E), device=dev)
        gran, xchg = ops.query_attn_rng(packed, h_t, Q, d[2], d[3], d[4], d[5], w2, c2, x_t, parts, epoch=epoch, granules=gran, xchg=xchg,
                                        status=status)
        x2 = ops.untile_rows(x_t, B, E)
        errs = dict(w_vs_split=maxdiff(w2, w1), ctx_vs_split=maxdiff(x2, x1), w=maxdiff(w2, w_r), ctx=maxdiff(x2, x_r),
                    cum=maxdiff(c2, w_r + w_cum.double()))
        report('query_attn_rng', B=B, L=L, parts=parts, epoch=epoch, **errs)
        assert errs['w_vs_split'] < 1e-6 and errs['ctx_vs_split'] < 1e-5
        assert errs['w'] < 2e-6 and errs['ctx'] < 2e-5 and errs['cum'] < 2e-6
        assert int(status.item()) == 0
    with pytest.raises(RuntimeError):       # E / parts must be whole float4 columns; 9 parts are not offered
        ops.query_attn_rng(packed, h_t, Q, d[2], d[3], d[4], d[5], w2, c2, x_t, 9, epoch=4, granules=gran)


def test_long_text_decode_uses_the_one_launch_form_and_matches_the_three_launch_form(dev):
    """Decoder.forward at L = 171 (free running): the loop with query projection + range fin part + combine as one launch per step
    == the loop with the split + combine launches (same split of the softmax; ~1e-7), both finite"""
    from helpers import full_tacotron
    from semi_tts_amd.synthetic import synthetic_batch
    B, L, T = 8, 171, 36
    m = full_tacotron(dev, seed=21, prenet_dropout=0.0)
    txt, spk, _ = synthetic_batch(B, L, T, seed=9)
    txt, spk = torch.from_numpy(txt).to(dev), torch.from_numpy(spk).to(dev)
    with torch.no_grad():
        mem = m.encoder(txt, None).contiguous()
        m.decoder.attn_rng_one_launch = True
        mel1, al1, _ = m.decoder(mem, None, T, spk, tf_rate=0.0)
        assert 'attn_xchg' in m.decoder.last_tapes
        m.decoder.attn_rng_one_launch = False
        mel2, al2, _ = m.decoder(mem, None, T, spk, tf_rate=0.0)
        assert 'attn_xchg' not in m.decoder.last_tapes
    errs = dict(mel=maxdiff(mel1, mel2), align=maxdiff(al1, al2))
    report('decode_long_text_one_launch', **errs)
    assert errs['mel'] < 5e-6 and errs['align'] < 1e-6 and bool(torch.isfinite(mel1).all())


def test_asr_postnet_against_reference_golden(dev):
    """ASRPostnet (2-layer BiLSTM -> Linear -> log_softmax, src/asr.py:67-80) in eval mode vs the output recorded from the reference"""
    from semi_tts_amd.asr import ASRPostnet
    W, A, meta = load_golden('asr_postnet_tiny')
    m = ASRPostnet(meta['latent_dim'], meta['vocab_size'])
    m.load_state_dict(W)
    m = m.to(dev).eval()
    with torch.no_grad():
        y = m(A['x'].to(dev))
    err = maxdiff(y, A['y'])
    report('asr_postnet_golden', err=err)
    assert y.shape == A['y'].shape and err < 2e-5
    assert float((y.exp().sum(-1) - 1).abs().max()) < 1e-5


def test_parameter_layouts_refresh_in_one_launch(dev):
    """ops._ParamLayouts: tap-major (forward) and tap-reversed transposed (input-gradient) forms of conv / linear weights are cached
    per weight version and equal torch's permute / flip copies; a version bump of ANY registered weight refreshes all of them."""
    from semi_tts_amd import ops
    ws = [rnd(24, 16, 5, seed=1).to(dev), rnd(128, 80, 3, seed=2).to(dev), rnd(40, 36, seed=3).to(dev), rnd(16, 20, 7, seed=4).to(dev)]
    def check():
        for w in ws:
            if w.dim() == 3:
                assert torch.equal(ops._tap_major(w), w.permute(0, 2, 1).contiguous())
            wt, tm = ops.dx_weight(w)
            if w.dim() == 2:
                assert not tm and torch.equal(wt, w.t().contiguous())
            elif tm:
                assert torch.equal(wt, w.permute(1, 0, 2).flip(2).permute(0, 2, 1).contiguous())
            else:
                assert torch.equal(wt, w.permute(1, 0, 2).flip(2).contiguous())
    check()
    n0 = ops._LAYOUTS.refreshes
    check()
    assert ops._LAYOUTS.refreshes == n0                       # nothing stale: no launch
    for w in ws:
        w.mul_(1.5)                                           # (in-place update = what the optimiser does)
    check()
    assert ops._LAYOUTS.refreshes == n0 + 1                   # one refresh served every entry
    # the conv through the cached layouts == torch's conv, forward and input gradient
    x = rnd(3, 11, 80, seed=9).to(dev)
    y = ops.gemm(x, ws[1], pad=1)
    ref = torch.nn.functional.conv1d(x.cpu().double().transpose(1, 2), ws[1].cpu().double(), padding=1).transpose(1, 2)
    assert maxdiff(y, ref) < 1e-4


def test_gemm_batch_equals_separate_launches(dev):
    """ops.gemm(collect=...) + ops.gemm_flush: several conv / linear jobs of different tap counts and widths in one launch
    (st_gemm_fwd_batch; the CBHG conv bank) write exactly what one st_gemm_fwd launch per job writes -- same k order per output
    element -- including an odd number of jobs beyond one launch (9 > 8) and a batch that falls back to separate launches."""
    from semi_tts_amd import ops
    x = rnd(5, 37, 80, seed=1).to(dev)
    ws = [rnd(80, 80, k, scale=(80 * k) ** -0.5, seed=10 + k).to(dev) for k in range(1, 10)]
    bs = [rnd(80, seed=30 + k).to(dev) for k in range(1, 10)]
    T = x.shape[1]
    ref = torch.zeros(5, T, 80 * 9, device=dev)
    got = torch.zeros_like(ref)
    for i, (w, b) in enumerate(zip(ws, bs)):
        ops.gemm(x, w, ref, pad=(i + 1) // 2, Tout=T, coff=80 * i, bias=b, act_pre='relu')
    jobs = []
    for i, (w, b) in enumerate(zip(ws, bs)):
        ops.gemm(x, w, got, pad=(i + 1) // 2, Tout=T, coff=80 * i, bias=b, act_pre='relu', collect=jobs)
    assert len(jobs) == 9 and float(got.abs().max()) == 0.0            # nothing ran yet
    ops.gemm_flush(jobs)
    assert jobs == [] and torch.equal(got, ref)
    cpu = torch.relu(torch.nn.functional.conv1d(x.cpu().double().transpose(1, 2), ws[2].cpu().double(), bs[2].cpu().double(), padding=1))
    assert maxdiff(got[:, :, 160:240], cpu.transpose(1, 2)) < 1e-4
    # a job the batched kernel cannot take (Cin not a multiple of 4 -> generic kernel): the flush runs the jobs one by one
    x2 = rnd(3, 20, 6, seed=2).to(dev)
    w2a, w2b = rnd(8, 6, seed=3).to(dev), rnd(12, 6, seed=4).to(dev)
    jobs = []
    ya = ops.gemm(x2, w2a, collect=jobs)
    yb = ops.gemm(x2, w2b, collect=jobs)
    ops.gemm_flush(jobs)
    assert torch.equal(ya, ops.gemm(x2, w2a)) and torch.equal(yb, ops.gemm(x2, w2b))


@pytest.mark.parametrize('M,C,NL', [(8256, 80, 4), (70, 80, 4), (33, 16, 1), (100, 64, 8)])
def test_highway_stack_in_one_launch_is_bitwise_the_layer_by_layer_form(dev, M, C, NL):
    """st_highway_stack_fwd (all highway layers of the CBHG in one launch, the rows of a workgroup kept in LDS) against the
    two-GEMMs-per-layer form (same k order per output element: bit-identical) and against torch in float64."""
    from semi_tts_amd import ops
    x = rnd(M, C, seed=1).to(dev)
    layers = [(rnd(C, C, scale=C ** -0.5, seed=10 + 4 * l).to(dev), rnd(C, seed=11 + 4 * l).to(dev),
               rnd(C, C, scale=C ** -0.5, seed=12 + 4 * l).to(dev), rnd(C, seed=13 + 4 * l).to(dev)) for l in range(NL)]
    y = ops.highway_stack(x, layers)
    assert y is not None
    ref, r64 = x, x.cpu().double()
    for wh, bh, wt, bt in layers:
        h = ops.gemm(ref, wh, bias=bh, act_pre='relu')
        ref = ops.gemm(ref, wt, bias=bt, act_pre='sigmoid', highway_h=h, res=ref)
        H = torch.relu(r64 @ wh.cpu().double().t() + bh.cpu().double())
        T = torch.sigmoid(r64 @ wt.cpu().double().t() + bt.cpu().double())
        r64 = H * T + r64 * (1.0 - T)
    assert torch.equal(y, ref)
    assert maxdiff(y, r64) < 1e-4
    assert ops.highway_stack(rnd(40, 96, seed=2).to(dev), [(rnd(96, 96, seed=3).to(dev), None, rnd(96, 96, seed=4).to(dev), None)]) is None
