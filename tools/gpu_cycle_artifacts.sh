#!/bin/bash
# MI355X box: the artifacts of the speech <-> text cycles (BASELINE config 3 as the reference trains it).  usage: bash tools/gpu_cycle_artifacts.sh r06 [quick]
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/cycle_$TAG
mkdir -p $OUT
cd $ROOT
echo "== main.py default mode (VqvaeTrainer) on config 3, 20 steps"
timeout 600 python main.py --config config/semi-single-spkr-paired-data.yaml --max-step 20 > $OUT/main_cycle_20_steps.log 2>&1; echo "exit $?"; tail -4 $OUT/main_cycle_20_steps.log
echo "== bench.py --workload cycle (B = 32 + 32, then the configuration file's 8 + 8)"
if [ "$2" == "quick" ]; then NOCPU=--no-cpu-baseline; fi
timeout 900 python bench.py --workload cycle --steps 20 --warmup 6 $NOCPU > $OUT/bench_cycle.json 2> $OUT/bench_cycle.err; echo "exit $?"; cut -c1-400 $OUT/bench_cycle.json
timeout 900 python bench.py --workload cycle --batch-size 8 --steps 20 --warmup 6 --no-cpu-baseline > $OUT/bench_cycle_b8.json 2> $OUT/bench_cycle_b8.err; echo "exit $?"; cut -c1-400 $OUT/bench_cycle_b8.json
echo "== kernel trace: one steady-state step of each cycle kind"
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace -d $OUT/prof -o cycle -- python3 $ROOT/bench.py --workload cycle --steps 2 --warmup 2 --no-cpu-baseline > $OUT/bench_cycle_under_rocprof.json 2> $OUT/prof.err)
DB=$(find $OUT/prof -name "*.db" | head -1)
python tools/prof_cycle_steps.py $DB $OUT/cycle > $OUT/cycle_one_step.txt 2>&1; head -80 $OUT/cycle_one_step.txt
rm -rf $OUT/prof
echo "== speech_to_text forward"
timeout 300 python tools/bench_speech_to_text.py > $OUT/bench_speech_to_text.json 2> $OUT/bench_speech_to_text.err; cat $OUT/bench_speech_to_text.json
