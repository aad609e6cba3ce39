#!/bin/bash
mkdir -p gpurun_out
for V in exact fast; do
  if [ $V = fast ]; then export ST_LIB_PATH=$PWD/tools/variants/libsemitts_vqfast.so; else unset ST_LIB_PATH; fi
  timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "vq" > gpurun_out/d_pytest_vq_$V.log 2>&1
  echo "pytest vq $V exit $?"; tail -n 2 gpurun_out/d_pytest_vq_$V.log
  timeout 600 python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/d_bench_c3_$V.json 2> gpurun_out/d_bench_c3_$V.err
  python - <<PY
import json
r=json.load(open('gpurun_out/d_bench_c3_$V.json'))
print('$V', [(c['vectors'], c['V'], c['us_per_launch']) for c in r['cases']])
PY
done
unset ST_LIB_PATH
timeout 2400 tools/gemm_lab2 v > gpurun_out/d_gemm_lab2.log 2>&1
echo "lab exit $?"; grep -E "^[a-z]|BEST" gpurun_out/d_gemm_lab2.log
