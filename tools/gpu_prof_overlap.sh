#!/bin/bash
# rocprofv3 kernel trace of the decode bench with the side-job mode (ST_OVERLAP=2); prints per-kernel stats
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/prof_ov2
cd /tmp && export TMPDIR=/tmp
export ST_OVERLAP=${1:-2}
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_ov2 -o bench -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/prof_ov2/bench.json 2> $ROOT/gpurun_out/prof_ov2/bench.err
cd $ROOT/gpurun_out/prof_ov2
for f in $(find . -name '*_results.db'); do python3 $ROOT/tools/prof_stats.py $f | head -12; done
find . -name "*.db" -size +30M -delete
