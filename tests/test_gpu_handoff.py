"""The in-launch hand-offs (decode loop: query projection -> attention fin part, 8-byte {value, tag} granules; the one-launch BiLSTM):
a failure is REPORTED -- a sticky device word the callers read (st_decoder_io.handoff_status, ops.handoff_starved / persist_starved) --
not only a NaN; the callers then fall back to the multi-launch form and run again; and the hand-off does not fail while other work
saturates the compute units.  Needs a real MI355X."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _granules(vals, tag, dev):
    """(A,) fp32 values -> A 64-bit {value, tag} words as a (2A,) int32 tensor (little endian: value word first)"""
    g = torch.empty(vals.numel(), 2, dtype=torch.int32)
    g[:, 0] = vals.view(torch.int32)
    g[:, 1] = tag
    return g.reshape(-1).to(dev)


def test_granule_wait_returns_values_or_reports_a_time_out():
    """st_handoff_wait_selftest runs the consumer routine of the fin workgroups (at_wait_granules): tags of this epoch -> the
    values, status untouched; one stale tag -> NaN AND bit 0 of the status word (after a bounded number of polls)"""
    from semi_tts_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device('cuda:0')
    A = 256
    vals = torch.randn(A)
    out = torch.zeros(A, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    g = _granules(vals, 7, dev)
    _lib.check(lib.st_handoff_wait_selftest(g.data_ptr(), 7, A, status.data_ptr(), out.data_ptr(), 1000, ops.stream_handle()), 'selftest')
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), vals) and int(status.item()) == 0
    ops.check_handoff(status)                               # nothing to report
    g2 = g.clone()
    g2[2 * 129 + 1] = 6                                     # one granule still carries the previous step's tag
    _lib.check(lib.st_handoff_wait_selftest(g2.data_ptr(), 7, A, status.data_ptr(), out.data_ptr(), 1000, ops.stream_handle()), 'selftest')
    torch.cuda.synchronize()
    assert bool(torch.isnan(out).all()) and int(status.item()) == 1
    with pytest.raises(RuntimeError, match='hand-off'):
        ops.check_handoff(status)
    assert int(status.item()) == 0                          # reading the word clears it


def test_decode_loop_hand_off_survives_saturated_compute_units():
    """hundreds of graph replays of the C2 decode loop (86 in-launch hand-offs each) while a second stream keeps every compute
    unit busy with large GEMMs: outputs stay bit-identical to the undisturbed replay and the status word stays clear"""
    from helpers import full_tacotron
    from semi_tts_amd.runtime import GraphedDecoder
    from semi_tts_amd.synthetic import synthetic_batch
    dev = torch.device('cuda:0')
    B, L, T = 32, 43, 258
    m = full_tacotron(dev, seed=11, prenet_dropout=0.5)
    txt, spk, _ = synthetic_batch(B, L, T, seed=5)
    txt, spk = torch.from_numpy(txt).to(dev), torch.from_numpy(spk).to(dev)
    with torch.no_grad():
        mem = m.encoder(txt, None).contiguous()
    gd = GraphedDecoder(m.decoder, B, L, T, dev).capture()
    ref = [t.clone() for t in gd(mem, spk, redraw=True)]
    torch.cuda.synchronize()
    from semi_tts_amd import ops
    assert not ops.handoff_starved(m.decoder.handoff_status)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=dev)
    bad = 0
    for i in range(300):
        with torch.cuda.stream(side):                       # ~1 ms of work on all 256 CUs per replay, overlapping it
            for _ in range(2):
                a2 = a @ a
        out = gd(redraw=False)
        if i % 25 == 24:
            torch.cuda.synchronize()
            bad += int(not all(torch.equal(x, y) for x, y in zip(out, ref)))
    torch.cuda.synchronize()
    del a2
    assert not ops.handoff_starved(m.decoder.handoff_status)        # no hand-off timed out under the load
    assert bad == 0 and bool(torch.isfinite(ref[0]).all())


def test_prenet_layer_two_inside_the_proj_launch_equals_the_separate_launch():
    """the rejected-but-kept form of the free-running step (Decoder.prenet2_in_proj, ST_P2=1: prenet layer 2 as consumer workgroups of
    the proj (+) gate (+) prenet-layer-1 launch, its operand handed over as granules) gives the outputs of the default form (one
    launch of its own) -- same k order per output element -- and leaves the status word clear"""
    from helpers import full_tacotron
    from semi_tts_amd import ops
    from semi_tts_amd.synthetic import synthetic_batch
    dev = torch.device('cuda:0')
    B, L, T = 32, 43, 60
    m = full_tacotron(dev, seed=12, prenet_dropout=0.5)
    txt, spk, _ = synthetic_batch(B, L, T, seed=6)
    txt, spk = torch.from_numpy(txt).to(dev), torch.from_numpy(spk).to(dev)
    outs = []
    with torch.no_grad():
        mem = m.encoder(txt, None).contiguous()
        for flag in (False, True):
            m.decoder.prenet2_in_proj = flag
            torch.manual_seed(3)                              # the same prenet dropout masks
            outs.append([t.clone() for t in m.decoder(mem, None, T, spk)])
    torch.cuda.synchronize()
    assert not ops.handoff_starved(m.decoder.handoff_status)
    for x, y in zip(*outs):
        assert bool(torch.isfinite(x).all()) and float((x - y).abs().max()) < 1e-6


def test_eager_forward_falls_back_to_two_launches_when_the_hand_off_was_starved():
    """Decoder.forward (eager inference) reads the hand-off status back: a word left set by a starved launch means the pass is NaN --
    the loop runs again as two launches per hand-off (bit-identical results), warns once, and keeps that form"""
    import warnings
    from helpers import full_tacotron
    from semi_tts_amd import ops
    from semi_tts_amd.synthetic import synthetic_batch
    dev = torch.device('cuda:0')
    m = full_tacotron(dev, seed=3, prenet_dropout=0.0)
    txt, spk, _ = synthetic_batch(4, 9, 12, seed=5)
    txt, spk = torch.from_numpy(txt).to(dev), torch.from_numpy(spk).to(dev)
    with torch.no_grad():
        mem = m.encoder(txt, None).contiguous()
        ref = m.decoder(mem, None, 12, spk, tf_rate=0.0)
        assert m.decoder.attn_pq_in_fin and m.decoder.handoff_status is not None and int(m.decoder.handoff_status.item()) == 0
        m.decoder.handoff_status.fill_(1)                   # what a timed-out wait leaves behind
        ops._DEGRADED.clear()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            out = m.decoder(mem, None, 12, spk, tf_rate=0.0)
        assert any('starved' in str(x.message) for x in w)
        assert not m.decoder.attn_pq_in_fin                 # the two-launch form stays for the rest of the process
        assert all(torch.equal(a, b) for a, b in zip(out, ref)) and bool(torch.isfinite(out[0]).all())
        out2 = m.decoder(mem, None, 12, spk, tf_rate=0.0)   # ... and the next forward is clean
        assert all(torch.equal(a, b) for a, b in zip(out2, ref))


def test_graph_replay_recovers_from_a_starved_hand_off():
    """GraphedDecoder.check(): a status word set by a replay (here: set by hand, as a starved launch would) -> the loop is captured
    again in its two-launch form and the call replayed; the outputs are the undisturbed ones bit for bit"""
    from helpers import full_tacotron
    from semi_tts_amd import ops
    from semi_tts_amd.runtime import GraphedDecoder
    from semi_tts_amd.synthetic import synthetic_batch
    dev = torch.device('cuda:0')
    B, L, T = 8, 11, 24
    m = full_tacotron(dev, seed=12, prenet_dropout=0.5)
    txt, spk, _ = synthetic_batch(B, L, T, seed=6)
    txt, spk = torch.from_numpy(txt).to(dev), torch.from_numpy(spk).to(dev)
    with torch.no_grad():
        mem = m.encoder(txt, None).contiguous()
    gd = GraphedDecoder(m.decoder, B, L, T, dev).capture()
    ref = [t.clone() for t in gd(mem, spk, redraw=True)]
    torch.cuda.synchronize()
    assert m.decoder.attn_pq_in_fin
    m.decoder.handoff_status.fill_(1)
    ops._DEGRADED.clear()
    with pytest.warns(RuntimeWarning, match='starved'):
        out = gd.check()
    assert not m.decoder.attn_pq_in_fin
    assert all(torch.equal(x, y) for x, y in zip(out, ref))
    out = gd(redraw=False)                                  # later replays use the new graph
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(out, ref))


def test_whole_forward_falls_back_when_the_one_launch_bilstm_was_starved():
    """Tacotron2.forward (eager inference): the one-launch BiLSTM's status word set -> the layer runs per step and the pass again"""
    from helpers import full_tacotron
    from semi_tts_amd import ops
    from semi_tts_amd.synthetic import synthetic_batch
    dev = torch.device('cuda:0')
    m = full_tacotron(dev, seed=4, prenet_dropout=0.0)
    txt, spk, _ = synthetic_batch(4, 9, 12, seed=7)
    txt, spk = torch.from_numpy(txt).to(dev), torch.from_numpy(spk).to(dev)
    old = ops.LSTM_PERSIST
    try:
        ops.LSTM_PERSIST = True
        with torch.no_grad():
            ref = m(txt, None, 12, spk, tf_rate=0.0)
            ops.persist_status(dev).fill_(2)                # what a starved layer leaves behind
            ops._DEGRADED.clear()
            with pytest.warns(RuntimeWarning, match='starved'):
                out = m(txt, None, 12, spk, tf_rate=0.0)
        assert not ops.LSTM_PERSIST
        for a, b in zip(out, ref):
            assert float((a - b).abs().max()) < 3e-6       # (the per-step form differs from the one-launch form by rounding only)
    finally:
        ops.LSTM_PERSIST = old
