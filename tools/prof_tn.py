#!/usr/bin/env python3
"""Development tool: the weight-gradient GEMM launches of the last training step in a rocprofv3 rocpd database, with their grids.
usage: tools/prof_tn.py <results.db>"""
import sqlite3
import sys
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
gx = [c for c in cols if 'grid' in c.lower()]
rows = con.execute("select name, start, end, %s from kernels order by start" % ', '.join(gx)).fetchall()
marks = [i for i, r in enumerate(rows) if 'mt_kernel<2>' in r[0]]
lo = marks[-6] if len(marks) >= 6 else 0
for r in rows[lo:marks[-1]]:
    if 'tn_kernel' in r[0] or 'tn_dma_kernel' in r[0]:
        print('%8.2f us  grid %s  %s' % ((r[2] - r[1]) / 1e3, r[3:], r[0][28:54]))
