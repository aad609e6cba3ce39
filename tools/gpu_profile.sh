#!/bin/bash
# Runs on the MI355X box: rocprofv3 kernel trace + stats of the default bench command.
# Output -> gpurun_out/prof_<tag>/ ; copy the *_kernel_stats.csv summary into profiles/ afterwards.
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$TAG -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/prof_$TAG/bench.json 2> $ROOT/gpurun_out/prof_$TAG/bench.err
echo "rocprof exit $?"
cd $ROOT/gpurun_out/prof_$TAG
find . -name "*.csv" | head
for f in $(find . -name "*kernel_stats.csv"); do echo "== $f"; head -30 $f; done
cat bench.json
# keep the merge-back small: the full kernel trace can be large
find . -name "*kernel_trace.csv" -size +20M -delete
