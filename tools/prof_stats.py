#!/usr/bin/env python3
"""Per-kernel statistics (count, avg/min/max duration, share) from a rocprofv3 rocpd database.
usage: tools/prof_stats.py gpurun_out/prof_<tag>/bench_results.db [--csv out.csv]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                       "from kernels group by name order by sum(end-start) desc").fetchall()
    tot = sum(r[5] for r in rows) or 1
    lines = ['name,calls,avg_us,min_us,max_us,total_us,percent']
    for r in rows:
        lines.append('"%s",%d,%.3f,%.3f,%.3f,%.1f,%.2f' % (r[0].replace('"', "'"), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                          r[5] / 1e3, 100.0 * r[5] / tot))
    out = '\n'.join(lines)
    if '--csv' in sys.argv:
        open(sys.argv[sys.argv.index('--csv') + 1], 'w').write(out + '\n')
    for r in rows[:14]:
        print('%-100s n=%6d avg=%8.2f us min=%7.2f max=%8.2f %5.1f%%' % (r[0][:100], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, 100 * r[5] / tot))


if __name__ == '__main__':
    main()
