#!/bin/bash
# Runs on the MI355X box: HBM traffic counters (separate --pmc passes, no tracing domains besides kernel-trace)
# usage: [PASSES="FETCH_SIZE WRITE_SIZE MFMA"] [PMC_ARGS="--workload c5"] [PMC_SFX=_c5] tools/gpu_pmc.sh <tag>
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for C in ${PASSES:-FETCH_SIZE WRITE_SIZE MFMA}; do
  D=$ROOT/gpurun_out/pmc_${TAG}${PMC_SFX}_$C
  mkdir -p $D
  CTRS=$C
  # matrix-core pass: busy cycles and issued F32 MFMA math ops (x512 = FLOP) next to the active-cycle count
  if [ $C = MFMA ]; then CTRS="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; fi
  timeout 900 rocprofv3 --pmc $CTRS --kernel-trace -d $D -o pmc -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${PMC_ARGS} > $D/bench.json 2> $D/bench.err
  echo "$C exit $?"
  ls -la $D
done
