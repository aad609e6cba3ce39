#!/bin/bash
# rocprofv3 kernel trace of the whole-forward bench (encoder + decoder + postnet), per-kernel stats
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-full}
mkdir -p $ROOT/gpurun_out/prof_$TAG
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$TAG -o full -- python3 $ROOT/tools/bench_full_forward.py > $ROOT/gpurun_out/prof_$TAG/bench.json 2> $ROOT/gpurun_out/prof_$TAG/bench.err)
DB=$(find $ROOT/gpurun_out/prof_$TAG -name "*.db" | head -1)
python $ROOT/tools/prof_stats.py $DB --csv $ROOT/gpurun_out/prof_$TAG/kernel_stats.csv | head -16
tail -1 $ROOT/gpurun_out/prof_$TAG/bench.json
find $ROOT/gpurun_out/prof_$TAG -name "*.db" -size +30M -delete
