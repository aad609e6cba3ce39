#!/bin/bash
# Runs on the MI355X box: HBM traffic counters (separate --pmc passes, no tracing domains besides kernel-trace)
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $ROOT/gpurun_out/pmc_${TAG}_$C
  timeout 900 rocprofv3 --pmc $C --kernel-trace -d $ROOT/gpurun_out/pmc_${TAG}_$C -o pmc -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/pmc_${TAG}_$C/bench.json 2> $ROOT/gpurun_out/pmc_${TAG}_$C/bench.err
  echo "$C exit $?"
  ls -la $ROOT/gpurun_out/pmc_${TAG}_$C
done
