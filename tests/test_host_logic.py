"""CPU-side tests: host logic of the drop-in modules, the C-ABI library's exported symbols, the
oracle at full size against the reference fixture, and the multi-process helpers (gloo).
No GPU needed; nothing here launches a kernel."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO, load_golden
from helpers import FULL_CFG, coin_source, full_hp


def test_library_exports_every_declared_symbol():
    """libsemitts_hip.so loads on a CPU-only box and exports every function include/semitts.h declares"""
    from semi_tts_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    hdr = open(os.path.join(REPO, 'include', 'semitts.h')).read()
    declared = sorted(set(re.findall(r'\b(st_[a-z0-9_]+)\s*\(', hdr)))
    assert len(declared) >= 40
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, 'declared in semitts.h but not exported: %s' % missing
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == declared
    # INTEGRATION.md quotes the count of entry points: keep it the header's
    quoted = re.search(r'entry points \((\d+)\)', open(os.path.join(REPO, 'INTEGRATION.md')).read())
    assert quoted and int(quoted.group(1)) == len(declared), (quoted and quoted.group(1), len(declared))
    lib2 = _lib.load()
    assert lib2.st_abi_version() == 1
    assert lib2.st_t16_floats(32, 240) == 2 * 15 * 256
    k = (ctypes.c_int * 3)(256, 512, 1024)
    assert lib2.st_packed_weight_floats(k, 3, 4096, 1024) == 256 * 112 * 256
    # where the library cuts the decoder cell's gate reduction (host arithmetic): half of [ctx | AdaIN(h_q) | h_d] in whole rounds of 16
    # k-blocks, never inside the context columns
    from semi_tts_amd._lib import StDecoderDims
    for (E, Q, D), want in (((512, 1024, 1024), 1280), ((512, 520, 520), None), ((64, 64, 64), None), ((512, 256, 256), None)):
        d = StDecoderDims(B=32, L=43, E=E, n_mels=80, r=3, P=256, Q=Q, D=D, A=128)
        cut = lib2.st_decoder_gate_split_k(ctypes.byref(d))
        kb = lambda n: (n + 15) // 16
        total = 16 * (kb(E) + kb(Q) + kb(D))
        assert cut % 16 == 0 and 16 * kb(E) <= cut < total, (E, Q, D, cut)
        assert want is None or cut == want


def test_product_path_refuses_cpu_tensors():
    from semi_tts_amd import ops
    from semi_tts_amd.tts import Tacotron2
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.linear_small(torch.zeros(2, 8), torch.zeros(4, 8))
    m = Tacotron2(80, 1025, 64, 128, json.loads(json.dumps(FULL_CFG))).eval()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 5, 64), None, 6, torch.zeros(1, 128))


def test_state_dict_keys_match_reference():
    """same parameter/buffer names and shapes as the reference modules, so its checkpoints load"""
    from semi_tts_amd.embed import L2Embedding, SeperateEmbedding
    from semi_tts_amd.tts import Tacotron2
    ref = json.load(open(os.path.join(GOLDEN, 'state_dict_keys.json')))
    m = Tacotron2(80, 1025, 64, 128, json.loads(json.dumps(FULL_CFG)))
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == ref['tts']
    W, _, _ = load_golden('vq_l2_native')
    attr = os.path.join(REPO, 'tests', 'golden', '_attr_tmp.csv')
    try:
        a = W['phn_attr.weight'].numpy()[3:]
        with open(attr, 'w') as f:
            f.write('\t' + '\t'.join('a%d' % i for i in range(a.shape[1])) + '\n')
            for i, row in enumerate(a):
                f.write('p%d\t' % i + '\t'.join(str(int(v)) for v in row) + '\n')
        kw = dict(softmax='normal', latent_dim=64, commit_weight=0, vq_weight=0, temp=1, skip_prob=0, stop_grad=True,
                  phn_attr_pth=attr, proj_attr=16)
        l2 = L2Embedding(43, False, **kw)
        sep = SeperateEmbedding(43, False, **kw)
    finally:
        os.remove(attr)
    assert {k: list(v.shape) for k, v in l2.state_dict().items()} == ref['l2']
    assert {k: list(v.shape) for k, v in sep.state_dict().items()} == ref['seperate']
    # the attribute table read from the tab-separated file equals the reference's (3 zero rows prepended)
    assert torch.equal(l2.phn_attr.weight, W['phn_attr.weight'])
    assert l2.out_dim == 64 and 'Temp.' in l2.create_msg()


@pytest.mark.parametrize('name', ['tts_tiny_infer', 'tts_tiny_train_tf', 'tts_tiny_sched', 'tts_tiny_partial', 'tts_tiny_quirk'])
def test_plan_decode_matches_reference_step_policy(name):
    """step count and next-input source sequence reproduce what the reference did (alignment length,
    number of host-RNG draws consumed)"""
    from semi_tts_amd.module import plan_decode
    _, A, meta = load_golden(name)
    hp = meta['hp']
    is_int = meta['teacher'] is not None
    B = A['txt_embed'].shape[0]
    Bt = B if is_int else A['teacher'].shape[0]
    draws = []
    src_coins = coin_source(A['coins'])

    def coin():
        draws.append(1)
        return src_coins()
    steps, src = plan_decode(is_int, meta['teacher'] if is_int else A['teacher'].shape[1], Bt, B,
                             hp['n_frames_per_step'], meta['tf_rate'], hp['drop_dec_in'], meta['unpair_max_frame'], coin)
    assert steps == A['align'].shape[1] and len(src) == steps
    assert len(draws) == len(A['coins'])
    if meta['tf_rate'] == 0.0:
        assert set(src) == {-1}
    if name == 'tts_tiny_train_tf':
        assert src == list(range(steps))
    if name == 'tts_tiny_sched':
        assert -1 in src and any(s >= 0 for s in src)


def test_synthetic_weights_are_deterministic_and_order_independent():
    from semi_tts_amd.synthetic import synthetic_batch, synthetic_state_dict
    shapes = {'a.weight': (8, 4), 'b.bias': (8,), 'c.running_var': (8,), 'd.num_batches_tracked': ()}
    s1 = synthetic_state_dict(shapes, 7)
    s2 = synthetic_state_dict(dict(reversed(list(shapes.items()))), 7)
    for k in shapes:
        assert np.array_equal(s1[k], s2[k])
    assert not np.array_equal(s1['a.weight'], synthetic_state_dict(shapes, 8)['a.weight'])
    assert (s1['c.running_var'] > 0).all() and s1['d.num_batches_tracked'].dtype == np.int64
    t1, t2 = synthetic_batch(2, 3, 6), synthetic_batch(2, 3, 6)
    assert all(np.array_equal(a, b) for a, b in zip(t1, t2))


def test_oracle_full_size_against_reference_fixture():
    """the oracle at FULL dimensions (B=4, T=66, L=12) with the seeded synthetic weights reproduces the
    real reference's output recorded in tests/golden/tts_full_c1.npz"""
    from oracle import tts_oracle as O
    from semi_tts_amd.synthetic import synthetic_state_dict
    _, A, meta = load_golden('tts_full_c1')
    W = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(meta['shapes'], meta['seed']).items()}
    torch.set_num_threads(8)
    with torch.no_grad():
        mel, lin, align, stop = O.tacotron2_forward(W, A['txt_embed'], meta['T'], A['spkr_embed'], full_hp(0.0))
    assert (mel - A['mel_infer']).abs().max() < 5e-5
    assert (align - A['align_infer']).abs().max() < 1e-5
    assert (lin[:, ::7, ::41] - A['linear_infer_s']).abs().max() < 2e-4


def test_shard_range_covers_everything_once():
    from semi_tts_amd.parallel import shard_range
    for n in (0, 1, 7, 32, 33, 64):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from semi_tts_amd.parallel import allreduce_gradients, broadcast_parameters, max_over_ranks, shard_range
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
torch.manual_seed(rank)                       # replicas start different ...
net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
broadcast_parameters(net)                     # ... and are made identical
x = torch.arange(8 * 6, dtype=torch.float32).view(8, 6) / 10.0
lo, hi = shard_range(8, world, rank)          # utterance-level sharding of one global batch
loss = net(x[lo:hi]).pow(2).sum() / 8.0
loss.backward()
n = allreduce_gradients(net.parameters(), bucket_bytes=64, average=False)
ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
ref.load_state_dict(net.state_dict())
(ref(x).pow(2).sum() / 8.0).backward()       # the single-process result on the whole batch
err = max((a.grad - b.grad).abs().max().item() for a, b in zip(net.parameters(), ref.parameters()))
t = max_over_ranks(1.0 + rank)
# the overlapped form: gradients accumulate straight into flat buckets, every bucket is all-reduced from an autograd hook as
# soon as its last gradient has arrived; a parameter that gets no gradient keeps .grad = None
from semi_tts_amd.parallel import GradReducer
net2 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
net2.load_state_dict({**net.state_dict(), '2.weight': torch.zeros(2, 3), '2.bias': torch.zeros(2)})
red = GradReducer(net2.parameters(), bucket_bytes=64, average=False)
for it in range(2):                                    # twice: the buckets are persistent and re-armed by prepare()
    red.prepare()
    (net2[1](net2[0](x[lo:hi])).pow(2).sum() / 8.0).backward()
    n2 = red.finish()
err2 = max((a.grad - b.grad).abs().max().item() for a, b in zip(list(net2.parameters())[:4], ref.parameters()))
unused_none = all(p.grad is None for p in net2[2].parameters())
views = all(p.grad.data_ptr() == red._view(p).data_ptr() for p in list(net2.parameters())[:4])
# ranks whose autograd graphs DIFFER (only rank 0 uses the third layer): the buckets still go out in index order on both ranks,
# and the parameter that fired on one rank only keeps the same summed gradient on BOTH (none of them drops it)
net3 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
broadcast_parameters(net3)
red3 = GradReducer(net3.parameters(), bucket_bytes=64, average=False)
red3.prepare()
h = net3[1](net3[0](x[lo:hi]))
(((net3[2](h) if rank == 0 else h).pow(2).sum()) / 8.0).backward()
red3.finish()
in_order = red3.order == sorted(red3.order) and len(red3.order) == len(red3.buckets)
g2 = torch.cat([p.grad.reshape(-1) if p.grad is not None else torch.full((p.numel(),), float('nan')) for p in net3[2].parameters()])
both = [torch.empty_like(g2) for _ in range(world)]
dist.all_gather(both, g2)
shared = bool(torch.equal(both[0], both[1]) and torch.isfinite(both[0]).all() and both[0].abs().max() > 0)
# static graph: after the first (recorded) pass only the hook of the parameter that completes each bucket is left
net4 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
net4.load_state_dict(net.state_dict())
red4 = GradReducer(net4.parameters(), bucket_bytes=64, average=False, static_graph=True)
for it in range(3):
    red4.prepare()
    (net4(x[lo:hi]).pow(2).sum() / 8.0).backward()
    n4 = red4.finish()
err4 = max((a.grad - b.grad).abs().max().item() for a, b in zip(net4.parameters(), ref.parameters()))
sparse = red4._sparse and len(red4._hooks) == len(red4.buckets) and n4 == len(red4.buckets)
# gradients BORN in their bucket slot: a producer that asks ops.grad_slot(param) writes there, autograd adopts the view (no copy,
# no accumulate); a producer that does not ask is gathered into its slot when the bucket goes out; nothing is zeroed
from semi_tts_amd import ops
class SlotLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w, b)
        return x @ w.t() + b
    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        dw, slot = dy.t() @ x, ops.grad_slot(w)
        if slot is not None:
            slot.copy_(dw)
            dw = slot
        return dy @ w, dw, dy.sum(0)                   # (the bias gradient does not ask: a stray tensor, gathered at launch)
net5 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
net5.load_state_dict(net.state_dict())
red5 = GradReducer(net5.parameters(), bucket_bytes=64, average=True, defer_average=True)
for it in range(2):
    red5.prepare()
    for flat in red5.flats:
        flat.fill_(float('nan'))                       # stale bucket contents must never show: every slot is written or zeroed
    h = SlotLinear.apply(x[lo:hi], net5[0].weight, net5[0].bias)
    (SlotLinear.apply(h, net5[1].weight, net5[1].bias).pow(2).sum() / 8.0).backward()
    red5.finish()
err5 = max((a.grad - b.grad).abs().max().item() for a, b in zip(net5.parameters(), ref.parameters()))   # SUMS over the ranks ...
born = (red5.stats['born_in_slot'] == 2 and red5.stats['gathered'] == 2 and red5.stats['zeroed'] == 0
        and red5.grad_scale == 1.0 / world                                                              # ... the 1 / world is the caller's
        and all(p.grad.data_ptr() == red5.ptr[p] for p in net5.parameters()))
red5.close()
no_sink = ops.grad_slot(net5[0].weight) is None
# the static-graph promise is CHECKED: when the graph changes after the recorded pass (a parameter that never fired starts to, and
# one that did stops), the step still reduces every gradient and the reducer goes back to per-parameter hooks with a warning
import warnings
net6 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3), torch.nn.Linear(5, 3))
broadcast_parameters(net6)
red6 = GradReducer(net6.parameters(), bucket_bytes=64, average=False, static_graph=True)
def run6(use_third):
    red6.prepare()
    h = net6[0](x[lo:hi])
    ((net6[2](h) if use_third else net6[1](h)).pow(2).sum() / 8.0).backward()
    red6.finish()
run6(False); run6(False); run6(False)
was_sparse = red6._sparse                              # (the ranks agree on the recorded order when the step after it begins)
run6(True)                                             # every rank sees the violation: this step's stray gradients are reduced at once
ref6 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3), torch.nn.Linear(5, 3))
ref6.load_state_dict(net6.state_dict())
(ref6[2](ref6[0](x)).pow(2).sum() / 8.0).backward()
err6 = max((a.grad - b.grad).abs().max().item() for a, b in zip(list(net6[0].parameters()) + list(net6[2].parameters()),
                                                                list(ref6[0].parameters()) + list(ref6[2].parameters())))
still_sparse = red6._sparse                            # ... and the layout changes when the NEXT step begins, on every rank alike
none1 = all(p.grad is None for p in net6[1].parameters())
with warnings.catch_warnings(record=True) as caught:
    warnings.simplefilter('always')
    run6(True)                                         # the per-parameter form carries on
reverted = (was_sparse and still_sparse and not red6._sparse and not red6.static_graph and len(red6._hooks) == 6
            and any('static_graph promise broken' in str(w.message) for w in caught) and none1)
err6 = max(err6, max((a.grad - b.grad).abs().max().item() for a, b in zip(net6[2].parameters(), ref6[2].parameters())))
# a violation on ONE rank only (advisor, round 5): the other rank must leave the static layout in the same step, or the next all-reduces
# differ in size and order (a hang).  Rank 1 alone starts using the third layer; both ranks revert when the next step begins.
net7 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3), torch.nn.Linear(5, 3))
broadcast_parameters(net7)
red7 = GradReducer(net7.parameters(), bucket_bytes=64, average=False, static_graph=True)
def run7(use_third):
    red7.prepare()
    h = net7[0](x[lo:hi])
    ((net7[2](h) if use_third else net7[1](h)).pow(2).sum() / 8.0).backward()
    red7.finish()
run7(False); run7(False); run7(False)
sparse7 = red7._sparse
with warnings.catch_warnings(record=True):
    warnings.simplefilter('always')
    run7(rank == 1)
    run7(False)                                        # (would hang or mix gradients if the two ranks' layouts differed)
    run7(False)
g7 = torch.cat([p.grad.reshape(-1) for p in list(net7[0].parameters()) + list(net7[1].parameters())])
both7 = [torch.empty_like(g7) for _ in range(world)]
dist.all_gather(both7, g7)
ref7 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3), torch.nn.Linear(5, 3))
ref7.load_state_dict(net7.state_dict())
(ref7[1](ref7[0](x)).pow(2).sum() / 8.0).backward()
r7 = torch.cat([p.grad.reshape(-1) for p in list(ref7[0].parameters()) + list(ref7[1].parameters())])
local7 = int(sparse7 and not red7._sparse and not red7.static_graph and bool(torch.equal(both7[0], both7[1]))
             and float((g7 - r7).abs().max()) < 1e-5)
if rank == 0:
    print('RESULT', err, n, t, err2, n2, int(unused_none), int(views), int(in_order), int(shared), err4, int(sparse),
          err5, int(born), int(no_sink), err6, int(reverted), local7)
dist.destroy_process_group()
'''


def test_two_process_gradient_allreduce_matches_single_process(tmp_path):
    """world_size 2 over gloo: shard a batch by utterance, all-reduce bucketed gradients, compare with
    the single-process gradient of the whole batch (the N>1 training path of BASELINE config 4)"""
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29531', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script), REPO], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0].splitlines() if l.startswith('RESULT')][0].split()
    assert float(line[1]) < 1e-5          # gradients equal the single-process ones
    assert int(line[2]) >= 2              # more than one bucket was exercised
    assert float(line[3]) == 2.0          # max over ranks of (1, 2)
    assert float(line[4]) < 1e-5 and int(line[5]) >= 2       # hook-driven buckets give the same gradients
    assert line[6] == '1' and line[7] == '1'                # unused parameters stay None, gradients are views of the buckets
    assert line[8] == '1'                                   # buckets issued in index order although the ranks' graphs differ
    assert line[9] == '1'                                   # a parameter that fired on one rank only: same gradient on both
    assert float(line[10]) < 1e-5 and line[11] == '1'       # static graph: one hook per bucket after the first pass, same gradients
    assert float(line[12]) < 1e-5 and line[13] == '1'       # gradients born in their bucket slots / gathered; sums + deferred 1 / world
    assert line[14] == '1'                                  # a closed reducer hands out no slots
    assert float(line[15]) < 1e-5 and line[16] == '1'       # broken static-graph promise: noticed, reduced correctly, reverted
    assert line[17] == '1'                                  # a violation on one rank only: both ranks leave the static layout in the same step


def test_cycle_step_alternates_and_gates_like_the_reference_loop():
    """VqvaeTrainer.cycle_step mirrors the loop body of bin/train_vqvae.py:124-150: even steps speech-first, odd steps text-first;
    the unpaired batch joins only when its weight is > 0 AND step > its start step."""
    from semi_tts_amd.solver import VqvaeTrainer
    tr = VqvaeTrainer.__new__(VqvaeTrainer)
    calls = []

    def sf(mel, aug, lin, text, sid, unpair_mel=None, unpair_aug_mel=None, unpair_linear=None, unpair_sid=None, _masks=None):
        calls.append(('speech', unpair_mel))
        tr.step += 1

    def tf(mel, aug, lin, text, sid, unpair_text=None, unpair_sid=None, _masks=None):
        calls.append(('text', unpair_text))
        tr.step += 1
    tr.speech_first_step, tr.text_first_step = sf, tf
    pair, unpair = ('m', 'a', 'l', 't', 's'), ('um', 'ua', 'ul', 'ut', 'us')
    tr.step = 0
    tr.hp = dict(unpair_text_weight=0.5, unpair_text_start_step=2, unpair_speech_weight=10.0, unpair_speech_start_step=1)
    for _ in range(6):
        tr.cycle_step(pair, unpair)
    #        step 0: speech, not past start; 1: text, not past 2; 2: speech + unpaired; 3: text + unpaired; ...
    assert calls == [('speech', None), ('text', None), ('speech', 'um'), ('text', 'ut'), ('speech', 'um'), ('text', 'ut')]
    calls.clear()
    tr.step = 4
    tr.hp = dict(unpair_text_weight=0.0, unpair_text_start_step=0, unpair_speech_weight=0.0, unpair_speech_start_step=0)
    tr.cycle_step(pair, unpair), tr.cycle_step(pair, unpair)
    assert calls == [('speech', None), ('text', None)]                   # zero weights (the shipped paired-only YAML): never joins
    calls.clear()
    tr.hp = dict(unpair_text_weight=1.0, unpair_speech_weight=1.0)
    tr.step = 5
    tr.cycle_step(pair, None)
    assert calls == [('text', None)]                                      # no unpaired batch handed in


@pytest.mark.parametrize('name', ['pretrained_l2', 'pretrained_seperate'])
def test_pretrained_partial_loads_match_reference(name, tmp_path):
    """VQVAE(pretrained_asr / _emb / _tts) with the reference's key-prefix rewriting (src/vqvae.py:70-90): the checkpoints and the
    state_dict of the model the REAL reference built from them are recorded in the golden (tools/gen_golden.py pretrained_case)"""
    import torch
    from conftest import load_golden
    from semi_tts_amd.vqvae import VQVAE
    W, A, meta = load_golden(name)
    ck = {'ckpt_asr': A['ckpt_asr'], 'ckpt_tts': A['ckpt_tts']}
    pa, pt = str(tmp_path / 'asr.pth'), str(tmp_path / 'tts.pth')
    torch.save({'model': ck['ckpt_asr']}, pa)
    torch.save({'model': ck['ckpt_tts']}, pt)
    cfg = dict(meta['model'])
    cfg['codebook'] = dict(cfg['codebook'])
    if cfg['codebook'].get('phn_attr_pth'):    # the attribute table as the tab-separated file the constructor reads (3 zero rows are prepended)
        a = W['codebook.phn_attr.weight'].numpy()[3:]
        attr = tmp_path / 'phn_attr.csv'
        attr.write_text('\t' + '\t'.join('a%d' % i for i in range(a.shape[1])) + '\n' +
                        ''.join('p%d\t' % i + '\t'.join(str(int(v)) for v in row) + '\n' for i, row in enumerate(a)))
        cfg['codebook']['phn_attr_pth'] = str(attr)
    cfg.update(pretrained_asr=pa, pretrained_tts=pt, pretrained_emb=pa if meta['bone'] == 'seperate' else '')
    torch.manual_seed(meta['seed'])
    m = VQVAE(8, 20, 43, 5, **cfg)
    assert [bool(m.pretrain_asr), bool(m.pretrained_emb), bool(m.pretrained_tts)] == meta['flags']
    sd = m.state_dict()
    assert set(sd.keys()) == set(W.keys())
    loaded = [k for k in W if k.startswith(('asr.', 'tts.decoder.', 'tts.postnet.')) or (meta['bone'] == 'seperate' and k == 'codebook.embedding.weight')]
    assert len(loaded) > 60
    for k in loaded:
        assert torch.equal(sd[k], W[k]), k
    # the text encoder is NOT part of the partial load (only decoder + postnet keys are taken from the TTS checkpoint)
    assert not torch.equal(sd['tts.encoder.convs.0.0.conv.weight'], ck['ckpt_tts']['encoder.convs.0.0.conv.weight'])
    if meta['bone'] == 'l2':                   # on the L2 codebook `embedding` is the table property: the reference's call fails, so does ours
        with pytest.raises(AttributeError):
            m.codebook.load_pretrained_embedding({'emb.weight': torch.zeros(43, 48)})


_RANKS_WORKER = r'''
import os, sys, json
from argparse import Namespace
sys.path.insert(0, sys.argv[1])
import bench
rk = bench.Ranks(Namespace(gpus=2, dist=False))
rk.barrier()
t = rk.max_seconds(1.0 + rk.rank)
info = rk.collectives_flat()
rk.close()
if rk.rank == 0:
    print('RESULT ' + json.dumps(dict(backend=rk.backend, t=t, **info)))
'''


@pytest.mark.parametrize('fake', [False, True])
def test_bench_ranks_fall_back_to_gloo_when_the_rccl_probe_fails(tmp_path, fake):
    """bench.py's rank set-up under `python -m torch.distributed.run` (how the driver starts N > 1): every rank probes RCCL in a
    CHILD process first; when the probe fails -- here genuinely (no GPU in this container) or faked (ST_BENCH_FAKE_RCCL_FAIL, what
    a failing peer-memory set-up looks like from outside) -- the process group comes up on gloo, the timing contract's barrier and
    max still work, and the line says rccl_ranks 0 with the probe's error."""
    script = tmp_path / 'worker.py'
    script.write_text(_RANKS_WORKER)
    env = dict(os.environ, ST_BENCH_ALLOW_CPU='1', ST_BENCH_PROBE_TIMEOUT='120')
    env.pop('ST_BENCH_BACKEND', None)
    if fake:
        env['ST_BENCH_FAKE_RCCL_FAIL'] = '1'
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29541' if fake else '29543', str(script), REPO], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=400)
    assert r.returncode == 0, r.stdout
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith('RESULT ')][0][7:])
    assert res['backend'] == 'gloo' and res['rccl_ranks'] == 0 and res['t'] == 2.0
    assert res['collectives']['backend'] == 'gloo' and res['collectives']['rccl_probe'].startswith('failed: rank')
    assert ('faked' in res['collectives']['rccl_probe']) == fake


def test_eight_rank_bring_up_agrees_on_gloo_when_one_rank_probe_fails(tmp_path):
    """The first 8-GPU run happens on a driver box with no chance to debug: eight ranks under torch.distributed.run, the RCCL probe
    passing on seven of them (faked: no GPU here) and failing on rank 3 only.  EVERY rank must come up on gloo (a group half on RCCL
    would hang), the barrier / max contract must work across all eight, and the line must name the rank whose probe failed."""
    script = tmp_path / 'worker.py'
    script.write_text(_RANKS_WORKER.replace('gpus=2', 'gpus=8'))
    env = dict(os.environ, ST_BENCH_ALLOW_CPU='1', ST_BENCH_PROBE_TIMEOUT='120', ST_BENCH_FAKE_RCCL_OK='1', ST_BENCH_FAKE_RCCL_FAIL='rank:3')
    env.pop('ST_BENCH_BACKEND', None)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
                        '--master-port', '29547', str(script), REPO], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=400)
    assert r.returncode == 0, r.stdout
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith('RESULT ')][0][7:])
    assert res['backend'] == 'gloo' and res['rccl_ranks'] == 0 and res['t'] == 8.0          # max over ranks of 1 + rank
    assert res['collectives']['rccl_probe'].startswith('failed: rank 3') and 'faked' in res['collectives']['rccl_probe']


_WORKER8 = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from semi_tts_amd import ops
from semi_tts_amd.parallel import GradReducer, broadcast_parameters, shard_range
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
sizes = [5, 4, 4, 4, 4, 4, 4, 3]                  # unequal shards of a global batch of 32 utterances
N = sum(sizes)
lo = sum(sizes[:rank]); hi = lo + sizes[rank]
spans = [shard_range(N, world, r) for r in range(world)]
cover = int(spans[0][0] == 0 and spans[-1][1] == N and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1)))
g = torch.Generator().manual_seed(0)
x = torch.randn(N, 6, generator=g)
class SlotLinear(torch.autograd.Function):        # a producer that writes its weight gradient where ops.grad_slot says (born in the bucket)
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w, b)
        return x @ w.t() + b
    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        dw, slot = dy.t() @ x, ops.grad_slot(w)
        if slot is not None:
            slot.copy_(dw); dw = slot
        return dy @ w, dw, dy.sum(0)
def build():
    torch.manual_seed(1234 + rank)              # replicas start different and are made identical
    net = torch.nn.Sequential(torch.nn.Linear(6, 7), torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    broadcast_parameters(net)
    return net
def fwd(net, rows):
    h = SlotLinear.apply(rows, net[0].weight, net[0].bias)
    h = torch.tanh(SlotLinear.apply(h, net[1].weight, net[1].bias))
    return net[2](h)
def ref_grads(net):
    ref = torch.nn.Sequential(torch.nn.Linear(6, 7), torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    ref.load_state_dict(net.state_dict())
    (fwd(ref, x).pow(2).sum() / N).backward()      # the single-process mean loss over the whole batch
    return [p.grad for p in ref.parameters()]
out = []
for static in (False, True):
    net = build()
    ref = ref_grads(net)
    red = GradReducer(net.parameters(), bucket_bytes=96, average=True, defer_average=True, static_graph=static)
    for it in range(4):
        red.prepare()
        # each rank's MEAN loss weighted by its share of the global batch (n_r * world / N): the average over ranks is the global mean
        (fwd(net, x[lo:hi]).pow(2).sum() / sizes[rank] * (sizes[rank] * world / N)).backward()
        nb = red.finish()
    scale = red.grad_scale                          # 1 / world: rides in the clip launch (optim.clip_grad_norm_(pre_scale=))
    err = max(float((p.grad * scale - r).abs().max()) for p, r in zip(net.parameters(), ref))
    norm = float(torch.sqrt(sum(((p.grad * scale).double() ** 2).sum() for p in net.parameters())))
    norm_ref = float(torch.sqrt(sum((r.double() ** 2).sum() for r in ref)))
    born = red.stats['born_in_slot']
    same = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(same, torch.tensor([float(sum(float(p.grad.sum()) for p in net.parameters()))]))
    agree = int(all(float(s) == float(same[0]) for s in same))
    out += [err, abs(norm - norm_ref) / norm_ref, int(scale == 1.0 / world), int(born >= 2), int(nb == len(red.buckets) and nb > 1), agree,
            int(red._sparse == static)]
    red.close()
if rank == 0:
    print('RESULT', cover, *out)
dist.destroy_process_group()
"""


def test_eight_process_reducer_with_unequal_shards_matches_single_process(tmp_path):
    """world_size 8 over gloo before any 8-GPU box sees the reducer: unequal utterance shards (5/4/4/4/4/4/4/3 of 32), gradients born
    in their bucket slots, the per-parameter (dynamic) and the one-hook-per-bucket (static, agreed across ranks) forms over four steps,
    the sums in the buckets with 1 / world left to the clip (grad_scale): averaged gradients and their norm = the single-process
    gradient of the whole batch; every rank ends with the same gradients bit for bit."""
    script = tmp_path / 'worker8.py'
    script.write_text(_WORKER8)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29551', WORLD_SIZE='8', OMP_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, str(script), REPO], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(8)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0].splitlines() if l.startswith('RESULT')][0].split()
    assert line[1] == '1'                                    # shard_range covers the batch once
    for form in range(2):                                    # dynamic, static
        err, nerr, scale_ok, born, buckets, agree, sparse = line[2 + 7 * form: 9 + 7 * form]
        assert float(err) < 1e-5 and float(nerr) < 1e-5, (form, err, nerr)
        assert scale_ok == '1' and born == '1' and buckets == '1' and agree == '1' and sparse == '1', (form, line)


# Command lines of the reference's README ("Running": train from scratch / continue / inference) and every flag its parser
# defines (ref: main.py:14-33).  Data, not code: an existing launch script must keep parsing.
_REFERENCE_COMMAND_LINES = [
    '--config config/supervised.yaml --njobs 8',
    '--config config/supervised.yaml --njobs 8 --load ckpt/x/step_10000.pth',
    '--gen-specgram --config config/supervised.yaml --njobs 8 --load ckpt/x/tts_1.pth --logdir out/',
    '--config config/semi-multi-spkr-paired-data.yaml --name exp --logdir log/ --ckpdir ckpt/ --seed 3 --njobs 5 --debug '
    '--no-pin --no-msg --actual-len --store-best-per --gen-wav',
    '--config config/supervised.yaml --cpu',
]


@pytest.mark.parametrize('line', _REFERENCE_COMMAND_LINES)
def test_main_parses_the_reference_command_lines(line):
    sys.path.insert(0, REPO)
    import main as entry
    paras = entry.parse_args(line.split())
    assert paras.gpu == (not paras.cpu) and paras.verbose == (not paras.no_msg)
    assert paras.pin_memory == (False if paras.cpu else paras.no_pin)     # inverted in the reference too (main.py:40)
    assert paras.config.startswith('config/')


@pytest.mark.parametrize('flag', ['--asr-decode', '--gen-gt-specgram', '--asr-only'])
def test_main_names_the_solvers_the_reference_tree_lacks(flag, capsys):
    sys.path.insert(0, REPO)
    import main as entry
    with pytest.raises(SystemExit):
        entry.parse_args(['--config', 'config/supervised.yaml', flag])
    assert 'not part of the reference tree' in capsys.readouterr().err
