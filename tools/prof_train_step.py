#!/usr/bin/env python3
"""One steady-state TRAINING step out of a rocprofv3 rocpd database: the launches between the last two fused-Adam launches
(the optimiser runs once per step), per kernel: launches, average and total duration; plus gaps (idle time between launches).
usage: tools/prof_train_step.py <results.db> [--csv out.csv] [--seq]"""
import collections
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if 'mt_kernel<2>' in r[0] or 'mt_tab_kernel<2>' in r[0]]       # the fused Adam update (optim.hip)
    if len(marks) < 2:
        sys.exit('fewer than two optimiser launches in the trace')
    # the optimiser may be several launches per step (chunks): group marks closer than 200 us
    groups = [[marks[0]]]
    for i in marks[1:]:
        if rows[i][1] - rows[groups[-1][-1]][2] < 200000:
            groups[-1].append(i)
        else:
            groups.append([i])
    lo, hi = groups[-2][-1] + 1, groups[-1][-1] + 1
    step = rows[lo:hi]
    wall = step[-1][2] - rows[lo - 1][2]
    busy = sum(r[2] - r[1] for r in step)
    agg = collections.OrderedDict()
    for name, s, e in step:
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += e - s
    out = sorted(agg.items(), key=lambda kv: -kv[1][1])
    print('steady-state step: %d launches, wall %.3f ms, sum of kernel time %.3f ms' % (len(step), wall / 1e6, busy / 1e6))
    lines = ['name,launches_per_step,avg_us,total_us,percent_of_wall']
    for name, (n, t) in out:
        lines.append('"%s",%d,%.3f,%.1f,%.2f' % (name.replace('"', "'"), n, t / n / 1e3, t / 1e3, 100.0 * t / wall))
    if '--csv' in sys.argv:
        open(sys.argv[sys.argv.index('--csv') + 1], 'w').write('\n'.join(lines) + '\n')
    for name, (n, t) in out[:45]:
        print('%-105s n=%5d avg=%8.2f us total=%8.1f us %5.2f%%' % (name[:105], n, t / n / 1e3, t / 1e3, 100.0 * t / wall))
    if '--seq' in sys.argv:
        prev = rows[lo - 1][2]
        for name, s, e in step:
            print('%9.2f gap %6.2f dur %8.2f  %s' % ((s - rows[lo - 1][2]) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, name[:100]))
            prev = e


if __name__ == '__main__':
    main()
