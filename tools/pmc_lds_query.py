#!/usr/bin/env python3
"""LDS bank-conflict ratio per kernel from a rocprofv3 --pmc database (tools/gpu_pmc_lds.sh); optional name filters."""
import sqlite3
import sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, counter_name, count(distinct dispatch_id), sum(counter_value) from pmc_events group by name, counter_name").fetchall()
per = {}
for name, cn, n, v in rows:
    per.setdefault(name, {})[cn] = v / max(n, 1)
    per[name]['n'] = n
for name, d in sorted(per.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', 0) * kv[1]['n']):
    if len(sys.argv) > 2 and not any(f in name for f in sys.argv[2:]):
        continue
    act = d.get('SQ_LDS_IDX_ACTIVE', 0) or 1
    print('%-90s n=%6d bank_conflict_cycles=%12.0f idx_active=%12.0f ratio=%.3f' % (name[:90], d['n'], d.get('SQ_LDS_BANK_CONFLICT', 0), act, d.get('SQ_LDS_BANK_CONFLICT', 0) / act))
