#!/bin/bash
# MI355X box: the whole-forward subset of tools/gpu_round_artifacts.sh (bench line with cpu_baseline, process-wide kernel stats, the
# one-launch BiLSTM against the per-step form).  usage: bash tools/gpu_forward_artifacts.sh r03
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/art_$TAG
mkdir -p $OUT
cd $ROOT
timeout 600 python tools/bench_full_forward.py 2>/dev/null | tail -1 > $OUT/bench_full_forward.json; cut -c1-300 $OUT/bench_full_forward.json
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_full -o full -- python3 $ROOT/tools/bench_full_forward.py --no-cpu-baseline > /dev/null 2> $OUT/prof_full.err)
DB=$(find $OUT/prof_full -name "*.db" | head -1); python tools/prof_stats.py $DB --csv $OUT/full_forward_kernel_stats.csv | head -10; rm -rf $OUT/prof_full
timeout 200 python tools/exp_lstm_persist.py 2>/dev/null | grep "^B=" > $OUT/bilstm_one_launch.txt; cat $OUT/bilstm_one_launch.txt
