#!/usr/bin/env python3
"""Summarise the HBM-traffic PMC passes (tools/gpu_pmc.sh) into profiles/<tag>_pmc_hbm_traffic<sfx>.{txt,json}.
usage: tools/pmc_summary.py <tag> [<sfx> <kernel name fragment> <algorithmic bytes per launch>]
       (reads gpurun_out/pmc_<tag><sfx>_{FETCH,WRITE}_SIZE/pmc_results.db; defaults: the C2 LSTM cell)
FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md (wide coalesced reads are counted at half their size); WRITE_SIZE as is."""
import json
import os
import sqlite3
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALGO_BYTES = 26050560          # mean algorithmic bytes of one LSTM-cell launch at C2 since round 6 (bench.py: lstm_probe -- query cell K = 1792,
                               # decoder cell K = 1280 + its 512 KB slab; 36356096 with ST_SPLIT_GATES=0: both cells at full K)


SFX = ''
KERNELS = ('pk_lstm_rt2_kernel', 'pk_kernel<0')


def _git_head():
    """the commit the profiled tree was built from: ST_COMMIT (the GPU box has no .git), else `git rev-parse` where this runs"""
    import subprocess
    try:
        return subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], cwd=os.path.dirname(os.path.abspath(__file__)), stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def stats(tag, counter):
    db = os.path.join(REPO, 'gpurun_out', 'pmc_%s%s_%s' % (tag, SFX, counter), 'pmc_results.db')
    con = sqlite3.connect(db)
    return con.execute("select name, count(*), avg(counter_value), min(counter_value), max(counter_value), avg(duration) "
                       "from pmc_events where counter_name = ? group by name order by sum(counter_value) desc limit 6",
                       (counter,)).fetchall()


def mfma(tag):
    """matrix-core pass -> profiles/<tag>_pmc_mfma.{txt,json}: issued F32 MFMA FLOP (MOPS x 512, counter_defs.yaml) per
    launch against the algorithmic 2*B*K*4H, FLOP/s against the 157.3 TFLOP/s fp32-matrix peak, busy cycles as counted"""
    db = os.path.join(REPO, 'gpurun_out', 'pmc_%s%s_MFMA' % (tag, SFX), 'pmc_results.db')
    if not os.path.exists(db):
        return
    con = sqlite3.connect(db)
    # the SQ counters come as one row per shader engine (32) and GRBM as one per XCD (8): sum (max for the active-cycle
    # count) over the rows of a dispatch first, then average over the dispatches of a kernel
    rows = con.execute("select name, counter_name, count(*), avg(v), avg(duration) from (select name, counter_name, dispatch_id, "
                       "duration, case when counter_name = 'GRBM_GUI_ACTIVE' then max(counter_value) else sum(counter_value) end as v "
                       "from pmc_events group by name, counter_name, dispatch_id) group by name, counter_name order by name").fetchall()
    per = {}
    for name, cn, n, avg, dur in rows:
        per.setdefault(name, dict(n=n, dur_ns=dur))[cn] = avg
    lines, out = [], {}
    for name, d in sorted(per.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0) * kv[1]['n']):
        flop = d.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0.0) * 512
        tfs = flop / max(d['dur_ns'], 1) / 1e3
        lines.append('%-70s n=%5d dur_ns=%9.0f mfma_flop=%14.0f TFLOP/s=%7.2f frac_of_157.3=%.3f busy_cycles=%.0f cu_busy=%.0f gui_active=%.0f'
                     % (name[:70], d['n'], d['dur_ns'], flop, tfs, tfs / 157.3, d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0),
                        d.get('SQ_BUSY_CU_CYCLES', 0), d.get('GRBM_GUI_ACTIVE', 0)))
        if any(k in name for k in KERNELS) and 'kernel' not in out:
            out = dict(kernel=name[:60], launches=d['n'], avg_duration_ns=round(d['dur_ns']), mfma_flop_per_launch=flop,
                       mfma_tflops=round(tfs, 2), frac_of_fp32_matrix_peak=round(tfs / 157.3, 4),
                       SQ_VALU_MFMA_BUSY_CYCLES=d.get('SQ_VALU_MFMA_BUSY_CYCLES'), GRBM_GUI_ACTIVE=d.get('GRBM_GUI_ACTIVE'),
                       SQ_BUSY_CU_CYCLES=d.get('SQ_BUSY_CU_CYCLES'),
                       mfma_util_pct_derived_formula=round(100 * d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) /
                                                           max(d.get('GRBM_GUI_ACTIVE', 1) * 1024, 1), 1),
                       note='MOPS x 512 = FLOP (counter_defs.yaml), equal to the algorithmic 2*B*K*4H; busy cycles = 32 per '
                            '16x16x4 f32 MFMA per SIMD; util = sum(busy) / (max GRBM_GUI_ACTIVE x 1024 SIMDs), the gfx94x '
                            'MfmaUtil formula; durations under PMC collection are ~7 % longer than in the plain kernel trace')
    open(os.path.join(REPO, 'profiles', '%s_pmc_mfma%s.txt' % (tag, SFX)), 'w').write('\n'.join(lines[:12]) + '\n')
    json.dump(out, open(os.path.join(REPO, 'profiles', '%s_pmc_mfma%s.json' % (tag, SFX)), 'w'), indent=1)
    print(json.dumps(out, indent=1))


def main():
    global SFX, KERNELS, ALGO_BYTES
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
    if len(sys.argv) > 4:
        SFX, KERNELS, ALGO_BYTES = sys.argv[2], (sys.argv[3],), int(float(sys.argv[4]))
    mfma(tag)
    if not os.path.exists(os.path.join(REPO, 'gpurun_out', 'pmc_%s%s_FETCH_SIZE' % (tag, SFX), 'pmc_results.db')):
        return
    lines, vals = [], {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        for name, n, avg, mn, mx, dur in stats(tag, c):
            lines.append('%-12s %-70s n=%5d avg=%12.1f min=%10.1f max=%12.1f avg_dur_ns=%s' % (c, name[:70], n, avg, mn, mx, dur))
            if any(k in name for k in KERNELS) and c not in vals:
                vals[c] = avg
                kname = name[:60]
    out = dict(kernel=kname, commit=os.environ.get('ST_COMMIT') or _git_head(), FETCH_SIZE_avg_KB=round(vals['FETCH_SIZE'], 1), WRITE_SIZE_avg_KB=round(vals['WRITE_SIZE'], 1),
               correction='MI355X_MICROARCH.md: FETCH_SIZE reads exactly 1/2 of wide coalesced reads on gfx950 -> x2; '
                          'WRITE_SIZE exact; both checked against kernels of known traffic in profiles/r05_pmc_calibration.txt',
               hbm_bytes_per_launch=int(round((2 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024)),
               algorithmic_bytes_per_launch=ALGO_BYTES,
               source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 2 --warmup 1 '
                      '--no-cpu-baseline%s`, see %s_pmc_hbm_traffic%s.txt' % (' ' + os.environ.get('PMC_ARGS', '') if SFX else '', tag, SFX))
    open(os.path.join(REPO, 'profiles', '%s_pmc_hbm_traffic%s.txt' % (tag, SFX)), 'w').write('\n'.join(lines) + '\n')
    json.dump(out, open(os.path.join(REPO, 'profiles', '%s_pmc_hbm_traffic%s.json' % (tag, SFX)), 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
