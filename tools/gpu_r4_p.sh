#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x > gpurun_out/p_pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -n 3 gpurun_out/p_pytest_gpu.log
timeout 600 python tools/bench_gemm_shapes.py 2>/dev/null | tail -1 > gpurun_out/p_gemm_shapes.json
python - <<PY
import json
for r in json.load(open('gpurun_out/p_gemm_shapes.json')): print('%-30s %7.2f us frac %.3f' % (r['shape'], r['us'], r['frac']))
PY
timeout 600 python tools/bench_full_forward.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/p_full_forward.json; cut -c1-420 gpurun_out/p_full_forward.json
timeout 900 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/p_train.json 2> gpurun_out/p_train.err; cut -c1-330 gpurun_out/p_train.json
