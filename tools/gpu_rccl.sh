#!/bin/bash
# Runs on the MI355X box: the RCCL branch on one GPU (world-size-1 nccl group, collectives forced).  Logs -> gpurun_out/
mkdir -p gpurun_out
echo "== rccl ws1 check"; timeout 900 python tools/rccl_ws1_check.py > gpurun_out/rccl_ws1.json 2> gpurun_out/rccl_ws1.err; echo "exit $?"; cut -c1-1500 gpurun_out/rccl_ws1.json; tail -n 5 gpurun_out/rccl_ws1.err
echo "== bench train --dist"; timeout 900 python bench.py --workload train --dist --steps 5 --warmup 2 > gpurun_out/bench_train_rccl_ws1.json 2> gpurun_out/bench_train_rccl_ws1.err; echo "exit $?"; cut -c1-1200 gpurun_out/bench_train_rccl_ws1.json; tail -n 5 gpurun_out/bench_train_rccl_ws1.err
echo "== bench c2 --dist"; timeout 600 python bench.py --dist --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c2_rccl_ws1.json 2> gpurun_out/bench_c2_rccl_ws1.err; echo "exit $?"; cut -c1-400 gpurun_out/bench_c2_rccl_ws1.json; tail -n 5 gpurun_out/bench_c2_rccl_ws1.err
