#!/bin/bash
# Runs on the MI355X box: HBM traffic counters (separate --pmc passes, no tracing domains besides kernel-trace)
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for C in ${PASSES:-FETCH_SIZE WRITE_SIZE MFMA}; do
  mkdir -p $ROOT/gpurun_out/pmc_${TAG}_$C
  CTRS=$C
  # matrix-core pass: busy cycles and issued F32 MFMA math ops (x512 = FLOP) next to the active-cycle count
  if [ $C = MFMA ]; then CTRS="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; fi
  timeout 900 rocprofv3 --pmc $CTRS --kernel-trace -d $ROOT/gpurun_out/pmc_${TAG}_$C -o pmc -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/pmc_${TAG}_$C/bench.json 2> $ROOT/gpurun_out/pmc_${TAG}_$C/bench.err
  echo "$C exit $?"
  ls -la $ROOT/gpurun_out/pmc_${TAG}_$C
done
