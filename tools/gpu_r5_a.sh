#!/bin/bash
# round 5, run A: gradient / DP tests after the born-in-bucket reducer, then the training step plain and through RCCL at world size 1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5a
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_gpu_grad.py tests/test_gpu_rccl.py tests/test_gpu_dp.py -q -x --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?"; tail -n 15 $OUT/pytest.log
timeout 600 python bench.py --workload train --steps 8 --warmup 3 --no-cpu-baseline 2>$OUT/bench_train.err | grep '^{"metric' | tail -1 > $OUT/bench_train.json; cut -c1-400 $OUT/bench_train.json
timeout 600 python bench.py --workload train --dist --steps 8 --warmup 3 --no-cpu-baseline 2>$OUT/bench_dist.err | grep '^{"metric' | tail -1 > $OUT/train_step_rccl_ws1.json
python - <<'PY'
import json,os
r=json.load(open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r5a/train_step_rccl_ws1.json'))
print({k:r[k] for k in ('ms_per_step','ms_variants','collectives_per_step','reducer') if k in r})
PY
tail -n 3 $OUT/bench_dist.err
