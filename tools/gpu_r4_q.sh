#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/q; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/p1 -o g -- python3 $ROOT/tools/bench_gemm_shapes.py > /dev/null 2> $OUT/p1.err)
DB=$(find $OUT/p1 -name "*.db" | head -1); python tools/prof_stats.py $DB | head -12; rm -rf $OUT/p1
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/p2 -o g -- $ROOT/tools/gemm_lab2 q > /dev/null 2> $OUT/p2.err)
DB=$(find $OUT/p2 -name "*.db" | head -1); python tools/prof_stats.py $DB | head -12; rm -rf $OUT/p2
