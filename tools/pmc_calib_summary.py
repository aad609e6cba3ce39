#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE per launch of tools/mb/mb_pmc_calib against the bytes each kernel is known to move.
usage: tools/pmc_calib_summary.py <dir with FETCH_SIZE/ and WRITE_SIZE/ rocprofv3 outputs>"""
import glob, os, sqlite3, sys
KNOWN = {'k_wide16': 256 << 20, 'k_row4': 2048 * 43 * 256 * 4, 'k_row4s': 2048 * 43 * 256 * 4, 'k_row16': 2048 * 43 * 512 * 4,
         'k_poll': 256 * 8 * 20000 * 8, 'k_write16': 256 << 20, 'k_write4': 256 << 20}
root = sys.argv[1]
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    dbs = glob.glob(os.path.join(root, ctr, '**', '*.db'), recursive=True)
    if not dbs:
        continue
    con = sqlite3.connect(dbs[0])
    rows = con.execute("select name, count(*), avg(v), min(v), max(v), avg(duration) from (select name, dispatch_id, duration, sum(counter_value) as v "
                       "from pmc_events where counter_name = ? group by name, dispatch_id) group by name", (ctr,)).fetchall()
    pair = con.execute("select dispatch_id, sum(counter_value) from pmc_events where counter_name = ? and name like 'k_pair%' group by dispatch_id "
                       "order by dispatch_id", (ctr,)).fetchall()
    if pair:
        print('   k_pair (two workgroups read one 44 KB tile at once; 128 tiles = %d KB unique): launches in order [same XCD, two XCDs, ...] reported KB: %s'
              % (128 * 43 * 256 * 4 // 1024, ' '.join('%.0f' % v for _, v in pair)))
    print('== %s (reported in KB per launch; summed over the rows of a dispatch)' % ctr)
    for name, n, avg, lo, hi, dur in rows:
        short = name.split('(')[0]
        if short not in KNOWN or (short == 'k_write16' and avg and avg > 6e5):      # (the 1 GiB flush launches are k_write16 too: listed apart)
            print('   %-10s n=%2d reported %12.0f KB (min %.0f max %.0f)   [flush / other]' % (short, n, avg, lo, hi))
            continue
        known = KNOWN[short]
        print('   %-10s n=%2d reported %12.0f KB  known %10.0f KB  known / reported = %.3f   dur %.1f us'
              % (short, n, avg, known / 1024.0, known / 1024.0 / max(avg, 1e-9), dur / 1e3))
