#!/bin/bash
# per-kernel times of the headline bench with prenet layer 2 inside the proj launch (ST_P2=1) and as a launch of its own (default)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for V in 0 1; do
  OUT=$ROOT/gpurun_out/p2_$V; mkdir -p $OUT
  export ST_P2=$((1-V))
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/prof.err)
  DB=$(find $OUT/prof -name "*.db" | head -1)
  python tools/prof_stats.py $DB | head -7 | cut -c1-170
  rm -rf $OUT/prof
done
