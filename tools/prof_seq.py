#!/usr/bin/env python3
"""Sequential listing of the LAST n kernel launches of a rocprofv3 rocpd database (start offset, gap, duration, name).
usage: tools/prof_seq.py <results.db> [n]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows = con.execute("select name, start, end from kernels order by start").fetchall()[-n:]
t0, prev = rows[0][1], rows[0][1]
for name, s, e in rows:
    print('%9.2f gap %7.2f dur %8.2f  %s' % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, name[:110]))
    prev = e
