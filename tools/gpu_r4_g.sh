#!/bin/bash
# kernel traces: one training step, the whole forward
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/gpu_train_prof.sh r04 > /dev/null 2>&1
cp gpurun_out/trainprof_r04/train_one_step_kernel_stats.csv gpurun_out/g_train_one_step_kernel_stats.csv
head -45 gpurun_out/trainprof_r04/train_one_step.txt
OUT=$ROOT/gpurun_out/g_full; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_full -o full -- python3 $ROOT/tools/bench_full_forward.py --no-cpu-baseline > /dev/null 2> $OUT/prof_full.err)
DB=$(find $OUT/prof_full -name "*.db" | head -1); python tools/prof_stats.py $DB --csv gpurun_out/g_full_forward_kernel_stats.csv | head -30; rm -rf $OUT/prof_full
