#!/bin/bash
# round 4, call A: VQ persistent loop (parity + c3 bench), GEMM lab, baseline GEMM shapes
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "vq" > gpurun_out/a_pytest_vq.log 2>&1
echo "pytest vq exit $?"; tail -n 5 gpurun_out/a_pytest_vq.log
timeout 600 python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/a_bench_c3.json 2> gpurun_out/a_bench_c3.err
echo "c3 exit $?"; cat gpurun_out/a_bench_c3.json | head -c 3000
timeout 600 python tools/bench_gemm_shapes.py > gpurun_out/a_gemm_shapes_base.log 2>&1
tail -n 15 gpurun_out/a_gemm_shapes_base.log | head -14
timeout 900 tools/gemm_lab2 v > gpurun_out/a_gemm_lab2.log 2>&1
echo "lab exit $?"; grep -E "^[a-z]|BEST" gpurun_out/a_gemm_lab2.log
