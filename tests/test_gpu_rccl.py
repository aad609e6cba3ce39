"""BASELINE config 4 as a tested configuration on ONE GPU: a world-size-1 RCCL (`nccl`) process group with every collective
forced drives the C2-size, 109-speaker training step through the hook-driven GradReducer (async all-reduce on RCCL's stream),
SyncBN (all_gather_into_tensor + all-reduce) and the max-over-ranks timing helper; the result must reproduce the plain step
(tools/rccl_ws1_check.py states the bars).  The check runs as a CHILD process: a failing RCCL bring-up then fails this test
instead of killing the pytest process.  ref: src/solver.py:138-151 (global-norm clip after the reduce)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    p = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'rccl_ws1_check.py')] + args, env=env, cwd=REPO,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert lines, 'no result line; rc=%d\n%s\n%s' % (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    return p.returncode, json.loads(lines[-1]), p.stderr


def test_c4_step_through_rccl_world_size_one_equals_plain_step():
    rc, res, err = _run([])
    report = os.path.join(REPO, 'gpurun_out')
    if os.path.isdir(report):
        with open(os.path.join(report, 'rccl_ws1.json'), 'w') as f:
            json.dump(res, f)
    assert res['backend'] == 'nccl' and res['rccl_ranks'] == 1
    ro, rs = res['reducer_only'], res['reducer_syncbn']
    assert ro['reducer_attached']
    assert ro['gradients_bitwise_equal'] and ro['updated_weights_bitwise_equal'], ro
    assert ro['loss'] == res['plain']['loss'] and ro['grad_norm'] == res['plain']['grad_norm']
    # third step, static-graph form: the big gradients are born in their bucket slots, the small rest is gathered, nothing is zeroed
    third = ro['third_step']['reducer']
    assert ro['third_step_loss_equal'] and third['sparse'] and third['zeroed'] == 0 and third['born_in_slot'] >= 40, ro
    assert rs['collectives_per_step'] == {'grad_buckets': 4, 'syncbn_fwd': 6, 'syncbn_bwd': 6, 'async_grad_buckets': 4}, rs
    assert rs['loss_abs_diff'] <= 1e-6 * max(1.0, abs(res['plain']['loss'])), rs
    assert rs['running_stats_max_rel_diff'] <= 1e-6, rs          # SyncBN-merged statistics: one rounding
    assert rs['grad_norm_rel_diff'] <= 1e-5, rs
    assert res['max_over_ranks'] == 1.25
    assert rc == 0 and res['ok'], (res, err[-2000:])


def test_cycle_steps_through_rccl_world_size_one_equal_the_plain_trainer():
    """config 4 as the reference trains it (VqvaeTrainer on the multi-speaker configuration): four alternating cycle steps through the
    dynamic GradReducer with RCCL all-reduces at world size 1 reproduce the plain trainer bit for bit; with SyncBN (12 BatchNorm gathers in
    a text-first step: speech encoder 6, text encoder 3, CBHG bank 1, projections 2) to one rounding."""
    rc, res, err = _run(['--cycle', '--batch-size', '8', '--frames', '64'])
    assert res['backend'] == 'nccl' and res['reducer_attached']
    assert res['statistics_equal'] and res['gradients_bitwise_equal'] and res['updated_weights_bitwise_equal'], res
    assert res['syncbn_collectives_last_step']['syncbn_fwd'] == 12 and res['syncbn_loss_max_rel_diff'] <= 1e-5, res
    assert rc == 0 and res['ok'], (res, err[-2000:])
