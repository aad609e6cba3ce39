#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/prof_c3
timeout 900 python -m pytest tests/test_gpu_grad.py tests/test_gpu_dp.py -m gpu -q --timeout 600 -p no:cacheprovider -k "speech_first or two_rank or vq_l2_backward or seperate" 2>&1 | tail -15
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "vq" 2>&1 | tail -3
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_c3 -o bench -- python3 $ROOT/bench.py --workload c3 --steps 10 > $ROOT/gpurun_out/prof_c3/bench.json 2> $ROOT/gpurun_out/prof_c3/bench.err)
DB=$(find $ROOT/gpurun_out/prof_c3 -name "*.db" | head -1)
python tools/prof_stats.py $DB --csv gpurun_out/prof_c3/kernel_stats.csv | head -4
timeout 300 python bench.py --workload c3 --steps 10 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
for c in r['cases']: print(c)"
ST_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --workload train --steps 3 --warmup 1 2>/dev/null | cut -c1-1500
