#!/bin/bash
# training step after a change: grad + rccl parity, bench (twice), one-step launch table
OUT=gpurun_out/r5f; mkdir -p $OUT
python -m pytest tests/test_gpu_grad.py tests/test_gpu_rccl.py tests/test_gpu_lstm_persist.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do
  python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $OUT/train_$i.json 2> $OUT/train_$i.err
  echo "train $(python -c "import json;r=json.load(open('$OUT/train_$i.json'));print(r['ms_per_step'])")"
done
bash tools/gpu_train_prof.sh r5f > /dev/null 2>&1; head -1 gpurun_out/trainprof_r5f/train_one_step.txt
grep -i "bwd_persist\|lstm_seq2_persist\|fillBuffer" gpurun_out/trainprof_r5f/train_one_step_kernel_stats.csv | cut -c1-150
