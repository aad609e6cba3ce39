#!/bin/bash
# MI355X box: the training-step subset of tools/gpu_round_artifacts.sh (bench line, RCCL world-size-1 line, kernel stats, one steady-state step).
# usage: bash tools/gpu_train_artifacts.sh r06
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/art_$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python bench.py --workload train --steps 30 --warmup 8 > $OUT/bench_train.json 2>/dev/null; cut -c1-300 $OUT/bench_train.json
timeout 900 python bench.py --workload train --dist --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | tail -1 > $OUT/train_step_rccl_ws1.json; cut -c1-200 $OUT/train_step_rccl_ws1.json
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_train -o bench -- python3 $ROOT/bench.py --workload train --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/prof_train.err)
DB=$(find $OUT/prof_train -name "*.db" | head -1)
python tools/prof_stats.py $DB --csv $OUT/train_kernel_stats.csv | head -12
rm -rf $OUT/prof_train
bash tools/gpu_train_prof.sh $TAG > /dev/null 2>&1; cp gpurun_out/trainprof_$TAG/train_one_step_kernel_stats.csv $OUT/train_one_step_kernel_stats.csv; head -1 gpurun_out/trainprof_$TAG/train_one_step.txt | tee $OUT/train_one_step_summary.txt
