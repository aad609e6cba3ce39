#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_full.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "fold or headline or c2_bench or edge_shapes or long_form or deterministic or attention" > gpurun_out/pytest_fold.log 2>&1; echo "pytest exit $?"; tail -n 25 gpurun_out/pytest_fold.log
timeout 600 python tools/debug_train_grads.py > gpurun_out/debug_train_grads.log 2>&1; echo "debug exit $?"; tail -n 32 gpurun_out/debug_train_grads.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"; cat gpurun_out/bench.json; tail -n 5 gpurun_out/bench.err
