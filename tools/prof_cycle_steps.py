#!/usr/bin/env python3
"""One steady-state step of EACH cycle kind out of a rocprofv3 rocpd database of `bench.py --workload cycle`: the trace is cut at the
fused-Adam launches (the optimiser runs once per step); the last segment that holds the run-length-merge kernel (vq_mean_fwd: only the
speech-first cycle with an unpaired batch runs it) and the last one that does not are written as per-kernel tables.
usage: tools/prof_cycle_steps.py <results.db> <out_prefix> [--seq]"""
import collections
import sqlite3
import sys


def table(rows, lo, hi, path, seq=False):
    step = rows[lo:hi]
    wall = step[-1][2] - rows[lo - 1][2]
    busy = sum(r[2] - r[1] for r in step)
    agg = collections.OrderedDict()
    for name, s, e in step:
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += e - s
    out = sorted(agg.items(), key=lambda kv: -kv[1][1])
    head = 'steady-state step: %d launches, wall %.3f ms, sum of kernel time %.3f ms' % (len(step), wall / 1e6, busy / 1e6)
    lines = ['name,launches_per_step,avg_us,total_us,percent_of_wall']
    for name, (n, t) in out:
        lines.append('"%s",%d,%.3f,%.1f,%.2f' % (name.replace('"', "'"), n, t / n / 1e3, t / 1e3, 100.0 * t / wall))
    open(path, 'w').write('\n'.join(lines) + '\n')
    print(path + ': ' + head)
    aten = [(n, c, t) for n, (c, t) in out if 'at::native' in n or 'at::cuda' in n or 'Kernel' in n and 'st_' not in n and 'elementwise' in n]
    print('  ATen launches: %d (%.1f us): %s' % (sum(c for _, c, _ in aten), sum(t for _, _, t in aten) / 1e3,
                                               '; '.join('%dx %s' % (c, n[:60]) for n, c, _ in aten[:12])))
    for name, (n, t) in out[:30]:
        print('  %-100s n=%5d avg=%8.2f us total=%8.1f us %5.2f%%' % (name[:100], n, t / n / 1e3, t / 1e3, 100.0 * t / wall))
    if seq:
        prev = rows[lo - 1][2]
        for name, s, e in step:
            print('%9.2f gap %6.2f dur %8.2f  %s' % ((s - rows[lo - 1][2]) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, name[:100]))
            prev = e
    return head


def main():
    con = sqlite3.connect(sys.argv[1])
    prefix = sys.argv[2]
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if 'mt_kernel<2>' in r[0] or 'mt_tab_kernel<2>' in r[0]]
    groups = []
    for i in marks:
        if groups and rows[i][1] - rows[groups[-1][-1]][2] < 200000:
            groups[-1].append(i)
        else:
            groups.append([i])
    segs = [(groups[j][-1] + 1, groups[j + 1][-1] + 1) for j in range(len(groups) - 1)]
    found = {}
    for lo, hi in segs:
        kind = 'speech_first' if any('vq_mean_fwd' in r[0] for r in rows[lo:hi]) else 'text_first'
        found[kind] = (lo, hi)
    for kind, (lo, hi) in found.items():
        table(rows, lo, hi, '%s_%s_one_step_kernel_stats.csv' % (prefix, kind), '--seq' in sys.argv)


if __name__ == '__main__':
    main()
