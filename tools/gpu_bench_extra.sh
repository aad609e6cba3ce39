#!/bin/bash
# Runs on the MI355X box: the secondary bench lines and the multi-rank launch paths of bench.py.  Logs -> gpurun_out/
mkdir -p gpurun_out
echo "== c3 (VQ)"; timeout 300 python bench.py --workload c3 --steps 10 --no-cpu-baseline > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err; echo "exit $?"; cat gpurun_out/bench_c3.json; tail -n 3 gpurun_out/bench_c3.err
echo "== train (C4 shape, 1 rank)"; timeout 600 python bench.py --workload train --steps 5 --warmup 2 > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; echo "exit $?"; cat gpurun_out/bench_train.json; tail -n 3 gpurun_out/bench_train.err
echo "== --gpus 2 on this box (must fail loudly when fewer than 2 GPUs)"; timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_g2.json 2> gpurun_out/bench_g2.err; echo "exit $?"; cat gpurun_out/bench_g2.json; tail -n 3 gpurun_out/bench_g2.err
echo "== 2 ranks sharing the GPU (gloo test hook): decode"; ST_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_g2_gloo.json 2> gpurun_out/bench_g2_gloo.err; echo "exit $?"; cat gpurun_out/bench_g2_gloo.json; tail -n 3 gpurun_out/bench_g2_gloo.err
echo "== 2 ranks sharing the GPU (gloo test hook): train"; ST_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --workload train --steps 3 --warmup 1 > gpurun_out/bench_train_g2_gloo.json 2> gpurun_out/bench_train_g2_gloo.err; echo "exit $?"; cat gpurun_out/bench_train_g2_gloo.json; tail -n 3 gpurun_out/bench_train_g2_gloo.err
echo "== c5"; timeout 300 python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; echo "exit $?"; cat gpurun_out/bench_c5.json; tail -n 3 gpurun_out/bench_c5.err
