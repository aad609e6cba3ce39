#!/usr/bin/env python3
"""Per-call-site kernel durations inside one decode step, from a rocprofv3 rocpd database."""
import collections
import sqlite3
import statistics as st
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
def code(n):
    if 'pk_kernel<0' in n or 'pk_lstm_rt2_kernel' in n: return 'L'
    if 'pk_kernel<1' in n: return 'l'
    if 'pk_attnpre_kernel' in n: return 'P'
    if 'pk_attnfin_kernel' in n: return 'F'
    if 'pk_attnfin_part_kernel' in n: return 'G'      # pq + fin with the decoder cell's hosted gate product (round 6)
    if 'at_kernel' in n: return 'A'
    return 'x'
seq = [(code(n), s, e) for n, s, e in rows]
text = ''.join(c for c, _, _ in seq)
for pat, names in (('LGLPl', ['LSTM_q', 'pq + attn fin part + hosted decoder-gate product (one launch)', 'LSTM_d (context columns + slab)', 'proj+pre0 (+ attn pre part of t+1)', 'pre1']),
                   ('LFLPl', ['LSTM_q', 'pq + attn fin part (one launch)', 'LSTM_d', 'proj+pre0 (+ attn pre part of t+1)', 'pre1']),
                   ('LlALPl', ['LSTM_q', 'pq', 'attn (fin part)', 'LSTM_d', 'proj+pre0 (+ attn pre part of t+1)', 'pre1']),
                   ('LlALll', ['LSTM_q', 'pq', 'attn', 'LSTM_d', 'proj+pre0', 'pre1']),
                   ('LlALlll', ['LSTM_q', 'pq', 'attn', 'LSTM_d', 'proj', 'pre0', 'pre1'])):
    dur = collections.defaultdict(list)
    i = 0
    n = len(pat)
    while i + n <= len(seq):
        if text[i:i + n] == pat and (i + n >= len(seq) or text[i + n] not in 'l'):
            for j in range(n):
                dur[j].append(seq[i + j][2] - seq[i + j][1])
            i += n
        else:
            i += 1
    if dur and len(dur[0]) > 50:
        tot = 0
        for j in range(n):
            m = st.median(dur[j]) / 1e3
            tot += m
            print('%-36s n=%5d median %6.2f us' % (names[j], len(dur[j]), m))
        print('sum of medians %.2f us' % tot)
        break
