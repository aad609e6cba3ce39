#!/usr/bin/env python3
"""Summarise the HBM-traffic PMC passes (tools/gpu_pmc.sh) into profiles/<tag>_pmc_hbm_traffic.{txt,json}.
usage: tools/pmc_summary.py <tag>   (reads gpurun_out/pmc_<tag>_{FETCH,WRITE}_SIZE/pmc_results.db)
FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md (wide coalesced reads are counted at half their size); WRITE_SIZE as is."""
import json
import os
import sqlite3
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALGO_BYTES = 36356096          # mean algorithmic bytes of one LSTM-cell launch at C2 (bench.py: lstm_algorithmic_bytes)


def stats(tag, counter):
    db = os.path.join(REPO, 'gpurun_out', 'pmc_%s_%s' % (tag, counter), 'pmc_results.db')
    con = sqlite3.connect(db)
    return con.execute("select name, count(*), avg(counter_value), min(counter_value), max(counter_value), avg(duration) "
                       "from pmc_events where counter_name = ? group by name order by sum(counter_value) desc limit 6",
                       (counter,)).fetchall()


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
    lines, vals = [], {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        for name, n, avg, mn, mx, dur in stats(tag, c):
            lines.append('%-12s %-70s n=%5d avg=%12.1f min=%10.1f max=%12.1f avg_dur_ns=%s' % (c, name[:70], n, avg, mn, mx, dur))
            if 'pk_kernel<0' in name:
                vals[c] = avg
    out = dict(kernel='pk_kernel<0,2,8,2>', FETCH_SIZE_avg_KB=round(vals['FETCH_SIZE'], 1), WRITE_SIZE_avg_KB=round(vals['WRITE_SIZE'], 1),
               correction='MI355X_MICROARCH.md: FETCH_SIZE reads exactly 1/2 of wide coalesced reads on gfx950 -> x2; '
                          'WRITE_SIZE uncalibrated, taken as is',
               hbm_bytes_per_launch=int(round((2 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024)),
               algorithmic_bytes_per_launch=ALGO_BYTES,
               source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 2 --warmup 1 '
                      '--no-cpu-baseline`, see %s_pmc_hbm_traffic.txt' % tag)
    open(os.path.join(REPO, 'profiles', '%s_pmc_hbm_traffic.txt' % tag), 'w').write('\n'.join(lines) + '\n')
    json.dump(out, open(os.path.join(REPO, 'profiles', '%s_pmc_hbm_traffic.json' % tag), 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
