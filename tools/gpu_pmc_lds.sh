#!/bin/bash
# Runs on the MI355X box: LDS bank-conflict counters per kernel (separate --pmc pass, kernel-trace only) for a command
# usage: tools/gpu_pmc_lds.sh <tag> <python script and args...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_lds_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --kernel-trace -d $OUT -o pmc -- python3 $ROOT/"$@" > $OUT/out.json 2> $OUT/out.err
echo "exit $?"
python3 - <<PY
import sqlite3
con = sqlite3.connect("$OUT/pmc_results.db")
rows = con.execute("select name, counter_name, count(distinct dispatch_id), sum(counter_value) from pmc_events group by name, counter_name").fetchall()
per = {}
for name, cn, n, v in rows:
    per.setdefault(name, {})[cn] = v / max(n, 1); per[name]['n'] = n
for name, d in sorted(per.items(), key=lambda kv: -kv[1].get('SQ_LDS_BANK_CONFLICT', 0) * kv[1]['n'])[:14]:
    act = d.get('SQ_LDS_IDX_ACTIVE', 0) or 1
    print('%-80s n=%6d bank_conflict_cycles=%12.0f idx_active=%12.0f ratio=%.3f' % (name[:80], d['n'], d.get('SQ_LDS_BANK_CONFLICT', 0), act, d.get('SQ_LDS_BANK_CONFLICT', 0) / act))
PY
find $OUT -name "*.db" -size +30M -delete
