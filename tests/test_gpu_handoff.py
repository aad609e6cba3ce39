"""The in-launch hand-off of the decode loop (query projection -> attention fin part, 8-byte {value, tag} granules): its failure
is an ERROR -- a sticky device word the callers check (st_decoder_io.handoff_status, ops.check_handoff) -- not only a NaN, and it
does not fail while other work saturates the compute units.  Needs a real MI355X."""
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
    gd.check()
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
    gd.check()                                              # raises if any hand-off timed out
    assert bad == 0 and bool(torch.isfinite(ref[0]).all())


def test_eager_forward_checks_the_status_word():
    """Decoder.forward (eager inference) reads the hand-off status back: a word left set by a starved launch raises"""
    from helpers import full_tacotron
    from semi_tts_amd.synthetic import synthetic_batch
    dev = torch.device('cuda:0')
    m = full_tacotron(dev, seed=3, prenet_dropout=0.0)
    txt, spk, _ = synthetic_batch(4, 9, 12, seed=5)
    txt, spk = torch.from_numpy(txt).to(dev), torch.from_numpy(spk).to(dev)
    with torch.no_grad():
        mem = m.encoder(txt, None).contiguous()
        m.decoder(mem, None, 12, spk, tf_rate=0.0)
        assert m.decoder.handoff_status is not None and int(m.decoder.handoff_status.item()) == 0
        m.decoder.handoff_status.fill_(1)                   # what a timed-out wait leaves behind
        with pytest.raises(RuntimeError, match='hand-off'):
            m.decoder(mem, None, 12, spk, tf_rate=0.0)
        m.decoder(mem, None, 12, spk, tf_rate=0.0)          # cleared by the report: the next forward is clean
