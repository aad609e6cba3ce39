#!/bin/bash
# full GPU suite, then the forward / training bench lines and the GEMM shape table after the load-serialisation fixes
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3
mkdir -p gpurun_out/s
timeout 600 python tools/bench_full_forward.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330
timeout 900 python bench.py --workload train --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-260
timeout 300 python tools/bench_gemm_shapes.py 2>/dev/null | tail -16
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-260
