#!/bin/bash
# Runs on the MI355X box: LDS bank-conflict counters per kernel (separate --pmc pass, kernel-trace only) for a command
# usage: [LDS_FILTER="name fragments"] tools/gpu_pmc_lds.sh <tag> <python script and args...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_lds_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --kernel-trace -d $OUT -o pmc -- python3 $ROOT/"$@" > $OUT/out.json 2> $OUT/out.err
echo "exit $?"
python3 $ROOT/tools/pmc_lds_query.py $OUT/pmc_results.db ${LDS_FILTER}
find $OUT -name "*.db" -size +30M -delete
