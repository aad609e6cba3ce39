#!/bin/bash
# C5 range-attention launch: query projection K-split over twice the workgroups (ST_RNG_KSPLIT=1) against the frozen form
OUT=gpurun_out/r5d; mkdir -p $OUT
for k in 0 1 0 1; do
  ST_RNG_KSPLIT=$k python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/c5_k$k.json 2> $OUT/c5_k$k.err
  echo "ksplit=$k $(python -c "import json;r=json.load(open('$OUT/c5_k$k.json'));print(r['value'], r['ms_per_step'])")"
done
ST_RNG_KSPLIT=1 python -m pytest tests -m gpu -q -x -k "c5 or handoff or long_form or range" 2>&1 | tail -3
python -m pytest tests -m gpu -q -x -k "c5 or handoff or long_form or range" 2>&1 | tail -3
