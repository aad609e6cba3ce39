#!/bin/bash
# MI355X box: the training step after a change -- gradient / RCCL / one-launch-LSTM parity, the train bench twice, the one-step launch table.  usage: bash tools/gpu_train_check.sh [tag]
TAG=${1:-chk}
OUT=gpurun_out/${TAG}; mkdir -p $OUT
python -m pytest tests/test_gpu_grad.py tests/test_gpu_rccl.py tests/test_gpu_lstm_persist.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do
  python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $OUT/train_$i.json 2> $OUT/train_$i.err
  echo "train $(python -c "import json;r=json.load(open('$OUT/train_$i.json'));print(r['ms_per_step'])")"
done
bash tools/gpu_train_prof.sh ${TAG} > /dev/null 2>&1; head -1 gpurun_out/trainprof_${TAG}/train_one_step.txt
