#!/bin/bash
# C5 experiment (round 4): the B = 64 LSTM cells as 128 workgroups of two row tiles x all four batch tiles (0.75 operand loads per MFMA)
export ST_LIB_PATH=$PWD/tools/variants/libsemitts_c5nb4.so
for v in 0 1; do
  if [ $v = 1 ]; then export ST_C5_NB4=1; else unset ST_C5_NB4; fi
  python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('NB4=$v', r['value'], 'frames/s', r['us_per_decode_step'], 'us/step', 'cell probe us', r['roofline']['avg_launch_us'])"
done
