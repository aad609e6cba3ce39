#!/bin/bash
# queued weight-gradient products (ops.flush_wgrads): parity, training step with and without, one-step launch table
OUT=gpurun_out/r5e; mkdir -p $OUT
python -m pytest tests/test_gpu_grad.py tests/test_gpu_rccl.py -m gpu -q -x 2>&1 | tail -4
for d in 0 1 0 1; do
  ST_WGRAD_DEFER=$d python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $OUT/train_d$d.json 2> $OUT/train_d$d.err
  echo "defer=$d $(python -c "import json;r=json.load(open('$OUT/train_d$d.json'));print(r['ms_per_step'])")"
done
bash tools/gpu_train_prof.sh r5e > /dev/null 2>&1; head -1 gpurun_out/trainprof_r5e/train_one_step.txt
