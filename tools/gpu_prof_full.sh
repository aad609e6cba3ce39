#!/bin/bash
# rocprofv3 kernel trace of the whole-forward bench (encoder + decoder + postnet), per-kernel stats
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/prof_full
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_full -o full -- python3 $ROOT/tools/bench_full_forward.py > $ROOT/gpurun_out/prof_full/bench.json 2> $ROOT/gpurun_out/prof_full/bench.err
cd $ROOT/gpurun_out/prof_full
for f in $(find . -name '*_results.db'); do python3 $ROOT/tools/prof_stats.py $f --csv full_kernel_stats.csv | head -24; done
find . -name "*.db" -size +30M -delete
