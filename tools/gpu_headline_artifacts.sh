#!/bin/bash
# MI355X box: the headline subset of tools/gpu_round_artifacts.sh (PMC passes of the C2 LSTM cell, headline bench with cpu_baseline, the
# kernel trace of the same command, the decode-step breakdown).  usage: bash tools/gpu_headline_artifacts.sh r06
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/art_$TAG
mkdir -p $OUT
cd $ROOT
export ST_COMMIT=${ST_COMMIT:-$(cat .st_commit 2>/dev/null)}
bash tools/gpu_pmc.sh $TAG > $OUT/pmc.log 2>&1; tail -3 $OUT/pmc.log
python tools/pmc_summary.py $TAG > $OUT/pmc_summary.log 2>&1; tail -5 $OUT/pmc_summary.log
cp profiles/${TAG}_pmc_hbm_traffic.* profiles/${TAG}_pmc_mfma.* $OUT/ 2>/dev/null; rm -rf $ROOT/gpurun_out/pmc_${TAG}*
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "exit $?"; cut -c1-300 $OUT/bench.json
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof.err)
DB=$(find $OUT/prof -name "*.db" | head -1)
python tools/prof_stats.py $DB --csv $OUT/bench_kernel_stats.csv | head -8
python tools/prof_steps.py $DB | tee $OUT/decode_step_breakdown.txt
rm -rf $OUT/prof
