#!/bin/bash
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "vq or speech_to_text or ctc" > gpurun_out/pytest_vq.log 2>&1; echo "pytest exit $?"; tail -n 15 gpurun_out/pytest_vq.log
timeout 300 python bench.py --workload c3 --steps 10 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err; echo "c3 exit $?"; cat gpurun_out/bench_c3.json; tail -n 3 gpurun_out/bench_c3.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fold > gpurun_out/bench_nofold.json 2> gpurun_out/bench_nofold.err; echo "nofold exit $?"; cut -c1-400 gpurun_out/bench_nofold.json
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_fold.json 2> gpurun_out/bench_fold.err; echo "fold exit $?"; cut -c1-400 gpurun_out/bench_fold.json
for TAG in fold nofold; do
  mkdir -p $ROOT/gpurun_out/prof_$TAG; EXTRA=""; [ $TAG = nofold ] && EXTRA="--no-fold"
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$TAG -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline $EXTRA > $ROOT/gpurun_out/prof_$TAG/bench.json 2> $ROOT/gpurun_out/prof_$TAG/bench.err)
  DB=$(find $ROOT/gpurun_out/prof_$TAG -name "*.db" | head -1)
  echo "== $TAG $DB"; python tools/prof_stats.py $DB --csv gpurun_out/prof_$TAG/kernel_stats.csv | head -8; python tools/prof_steps.py $DB
  find $ROOT/gpurun_out/prof_$TAG -name "*.db" -size +30M -delete
done
timeout 600 python tools/debug_train_grads.py > gpurun_out/debug_train_grads.log 2>&1; echo "debug exit $?"; tail -n 10 gpurun_out/debug_train_grads.log
