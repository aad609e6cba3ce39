#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x > gpurun_out/f_pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -n 15 gpurun_out/f_pytest_gpu.log
timeout 600 python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/f_bench_c3.json 2> gpurun_out/f_bench_c3.err
python - <<PY
import json
r=json.load(open('gpurun_out/f_bench_c3.json'))
print('c3', [(c['vectors'], c['V'], c['us_per_launch']) for c in r['cases']])
PY
ST_LIB_PATH=$PWD/tools/variants/libsemitts_vqstamps.so timeout 300 python tools/exp_vq_stamps.py 2>&1 | grep -v amdgpu.ids | head -14
timeout 600 python tools/bench_gemm_shapes.py 2>/dev/null | tail -1 > gpurun_out/f_gemm_shapes.json
python - <<PY
import json
for r in json.load(open('gpurun_out/f_gemm_shapes.json')): print('%-30s %7.2f us frac %.3f' % (r['shape'], r['us'], r['frac']))
PY
timeout 600 python tools/bench_full_forward.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/f_full_forward.json; cut -c1-600 gpurun_out/f_full_forward.json
timeout 900 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/f_train.json 2> gpurun_out/f_train.err; cut -c1-400 gpurun_out/f_train.json; tail -3 gpurun_out/f_train.err
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err; cut -c1-300 gpurun_out/f_bench.json
