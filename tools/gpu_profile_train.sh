#!/bin/bash
# Runs on the MI355X box: the training-step bench, then the same under rocprofv3 (kernel trace + stats).
TAG=${1:-r01_train}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/prof_$TAG
cd $ROOT
timeout 600 python3 tools/bench_train.py --steps 5 --warmup 2 > gpurun_out/prof_$TAG/train_bench.json 2> gpurun_out/prof_$TAG/train_bench.err
echo "bench exit $?"; cat gpurun_out/prof_$TAG/train_bench.json; tail -5 gpurun_out/prof_$TAG/train_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$TAG -o train -- python3 $ROOT/tools/bench_train.py --steps 2 --warmup 1 > $ROOT/gpurun_out/prof_$TAG/train_prof.json 2> $ROOT/gpurun_out/prof_$TAG/train_prof.err
echo "rocprof exit $?"
cd $ROOT/gpurun_out/prof_$TAG
for f in $(find . -name '*_results.db'); do python3 $ROOT/tools/prof_stats.py $f --csv train_kernel_stats.csv | head -40; done
find . -name "*kernel_trace.csv" -size +20M -delete
find . -name "*.db" -size +40M -delete
ls -la
