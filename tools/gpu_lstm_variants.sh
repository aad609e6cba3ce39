#!/bin/bash
# Timing experiment (MI355X box): LSTM-cell launch geometry (waves per workgroup x k-blocks per wave and group).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/variants
IFS=';' read -ra VARS <<< "${VARIANTS:--DPK_LSTM_KW=8 -DPK_LSTM_TRIP=2;-DPK_LSTM_KW=4 -DPK_LSTM_TRIP=4;-DPK_LSTM_KW=4 -DPK_LSTM_TRIP=3;-DPK_LSTM_KW=4 -DPK_LSTM_TRIP=2;-DPK_LSTM_KW=8 -DPK_LSTM_TRIP=3;-DPK_LSTM_KW=8 -DPK_LSTM_TRIP=1;-DPK_LSTM_KW=16 -DPK_LSTM_TRIP=1}"
i=0
for V in "${VARS[@]}"; do
  D=$ROOT/gpurun_out/variants/v$i; mkdir -p $D; OBJS=""
  for f in $ROOT/semi_tts_amd/csrc/*.hip; do
    o=$D/$(basename $f .hip).o
    if [ $(basename $f) = skinny_packed.hip ] || [ $(basename $f) = decoder.hip ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=16 $V -c $f -o $o || exit 1
    else
      o=$ROOT/semi_tts_amd/lib/$(basename $f .hip).o
    fi
    OBJS="$OBJS $o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib.so $OBJS || exit 1
  echo "== $V"
  ST_LIB_PATH=$D/lib.so timeout 300 python $ROOT/tools/exp_lstm.py 2>/dev/null
  for rep in 1 2; do
  ST_LIB_PATH=$D/lib.so timeout 300 python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS} 2> /dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('   bench us_per_step %.2f  lstm probe %.2f us' % (r['us_per_decode_step'], r['roofline']['avg_launch_us']))"
  done
  rm -rf $D
  i=$((i+1))
done
