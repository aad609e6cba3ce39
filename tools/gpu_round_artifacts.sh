#!/bin/bash
# Runs on the MI355X box: everything the round's profiles/ files are made from.  usage: bash tools/gpu_round_artifacts.sh r03
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/art_$TAG
mkdir -p $OUT
cd $ROOT
export ST_COMMIT=${ST_COMMIT:-$(cat .st_commit 2>/dev/null)}
echo "== PMC passes (HBM traffic, MFMA): C2 LSTM cell, C5 LSTM cell, VQ search"
bash tools/gpu_pmc.sh $TAG > $OUT/pmc.log 2>&1; tail -3 $OUT/pmc.log
python tools/pmc_summary.py $TAG > $OUT/pmc_summary.log 2>&1
PMC_ARGS="--workload c5" PMC_SFX=_c5 bash tools/gpu_pmc.sh $TAG > $OUT/pmc_c5.log 2>&1
PMC_ARGS="--workload c5" python tools/pmc_summary.py $TAG _c5 pk_lstm_rt2_kernel 49946624 >> $OUT/pmc_summary.log 2>&1    # mean of 4(4H K + 8H + B K + 3 B H), B = 64
PMC_ARGS="--workload c3 --vq-head-only" PMC_SFX=_c3 bash tools/gpu_pmc.sh $TAG > $OUT/pmc_c3.log 2>&1
PMC_ARGS="--workload c3 --vq-head-only" python tools/pmc_summary.py $TAG _c3 vq_l2_mfma_kernel 10731776 >> $OUT/pmc_summary.log 2>&1     # 4128 (520 + 4 x 512) + 512 x 64 x 4
cp profiles/${TAG}_pmc_* $OUT/ 2>/dev/null; rm -rf $ROOT/gpurun_out/pmc_${TAG}*
tail -5 $OUT/pmc_summary.log
echo "== headline bench"; timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "exit $?"; cut -c1-300 $OUT/bench.json
echo "== kernel trace of the same command"
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof.err)
DB=$(find $OUT/prof -name "*.db" | head -1)
python tools/prof_stats.py $DB --csv $OUT/bench_kernel_stats.csv | head -8
python tools/prof_steps.py $DB | tee $OUT/decode_step_breakdown.txt
rm -rf $OUT/prof
echo "== secondary benches (each with roofline + cpu_baseline)"
timeout 600 python bench.py --workload c5 --steps 5 --warmup 2 > $OUT/bench_c5.json 2>/dev/null; cut -c1-200 $OUT/bench_c5.json
timeout 300 python bench.py --workload c3 --steps 10 > $OUT/bench_c3.json 2>/dev/null; cut -c1-200 $OUT/bench_c3.json
timeout 900 python bench.py --workload train --steps 20 --warmup 5 > $OUT/bench_train.json 2>/dev/null; cut -c1-300 $OUT/bench_train.json
timeout 900 python bench.py --workload train --dist --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | tail -1 > $OUT/train_step_rccl_ws1.json; cut -c1-200 $OUT/train_step_rccl_ws1.json
timeout 600 python tools/rccl_ws1_check.py 2>/dev/null | grep '^{' | tail -1 > $OUT/rccl_ws1_check.json; cut -c1-200 $OUT/rccl_ws1_check.json
ST_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --workload train --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $OUT/train_step_2ranks_gloo_shared_gpu.json; cut -c1-200 $OUT/train_step_2ranks_gloo_shared_gpu.json
timeout 600 python tools/bench_full_forward.py 2>/dev/null | tail -1 > $OUT/bench_full_forward.json; cut -c1-300 $OUT/bench_full_forward.json
echo "== profiles of the secondary benches"
for W in c3 c5 train; do
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$W -o bench -- python3 $ROOT/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/prof_$W.err)
  DB=$(find $OUT/prof_$W -name "*.db" | head -1)
  python tools/prof_stats.py $DB --csv $OUT/${W}_kernel_stats.csv | head -12
  rm -rf $OUT/prof_$W
done
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_full -o full -- python3 $ROOT/tools/bench_full_forward.py --no-cpu-baseline > /dev/null 2> $OUT/prof_full.err)
DB=$(find $OUT/prof_full -name "*.db" | head -1); python tools/prof_stats.py $DB --csv $OUT/full_forward_kernel_stats.csv | head -10; rm -rf $OUT/prof_full
echo "== in-situ cost of each launch of the decode step (ablation build)"; bash tools/gpu_ablate.sh 2>&1 | grep "skip=" | tee $OUT/decode_step_ablation.txt
ABL_ARGS="--workload c5" SKIPS="0 1 6 8 16 32 64" bash tools/gpu_ablate.sh 2>&1 | grep "skip=" | tee $OUT/decode_step_ablation_c5.txt
echo "== one steady-state training step (launches and kernel time per step), GEMM shapes of the forward"
bash tools/gpu_train_prof.sh $TAG > /dev/null 2>&1; cp gpurun_out/trainprof_$TAG/train_one_step_kernel_stats.csv $OUT/train_one_step_kernel_stats.csv; head -1 gpurun_out/trainprof_$TAG/train_one_step.txt | tee $OUT/train_one_step_summary.txt
timeout 300 python tools/bench_gemm_shapes.py 2>/dev/null | tail -1 > $OUT/gemm_shapes.json; cut -c1-200 $OUT/gemm_shapes.json
MB_S=1 timeout 60 tools/mb/mb_attn_bwd 2>&1 | head -18 > $OUT/attn_bwd_phase_stamps.txt; head -1 $OUT/attn_bwd_phase_stamps.txt
