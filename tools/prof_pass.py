#!/usr/bin/env python3
"""One decode pass out of a rocprofv3 rocpd database of bench.py: the launches from the last but one `zero_regions_kernel` (first
launch of Decoder.forward) to the last one; prints every launch that is not one of the per-step kernels, and the totals.
usage: tools/prof_pass.py <results.db>"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if 'zero_regions_kernel' in r[0]]
lo, hi = marks[-2], marks[-1]
step_kernels = ('pk_lstm_rt2_kernel', 'pk_attnfin_kernel', 'pk_attnpre_kernel', 'pk_kernel<1, 1, 8, 2>', 'pk_attnrng_kernel')
t0 = rows[lo][1]
other = loop = 0.0
prev = rows[lo - 1][2]
for name, s, e in rows[lo:hi]:
    if any(k in name for k in step_kernels):
        loop += (e - s) / 1e3
    else:
        other += (e - s) / 1e3
        print('%9.2f gap %6.2f dur %7.2f  %s' % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, name[:120]))
    prev = e
print('pass: %d launches, wall %.1f us; per-step kernels %.1f us, everything else %.1f us' % (hi - lo, (rows[hi][1] - t0) / 1e3, loop, other))
