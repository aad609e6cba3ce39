#!/bin/bash
# MI355X box: kernel trace of the C2 training step, reduced to ONE steady-state step.  usage: bash tools/gpu_train_prof.sh [tag] [--seq]
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trainprof_$TAG
mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace -d $OUT/prof -o train -- python3 $ROOT/bench.py --workload train --steps 3 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/prof.err)
DB=$(find $OUT/prof -name "*.db" | head -1)
python $ROOT/tools/prof_train_step.py $DB --csv $OUT/train_one_step_kernel_stats.csv $2 > $OUT/train_one_step.txt
head -60 $OUT/train_one_step.txt
python $ROOT/tools/prof_tn.py $DB > $OUT/tn_launches.txt 2>&1
rm -rf $OUT/prof
