#!/bin/bash
# MI355X box: kernel trace of ONE steady-state training step through RCCL at world size 1 (every collective forced), beside the plain one:
# what data parallelism adds per kernel.  usage: bash tools/gpu_train_prof_ws1.sh [tag]
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trainprof_ws1_$TAG
mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace -d $OUT/prof -o train -- python3 $ROOT/bench.py --workload train --dist --no-variants --no-cpu-baseline --steps 3 --warmup 2 > $OUT/bench.json 2> $OUT/prof.err)
DB=$(find $OUT/prof -name "*.db" | head -1)
python $ROOT/tools/prof_train_step.py $DB --csv $OUT/train_ws1_one_step_kernel_stats.csv --seq > $OUT/train_ws1_one_step.txt
head -70 $OUT/train_ws1_one_step.txt
rm -rf $OUT/prof
