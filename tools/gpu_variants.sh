#!/bin/bash
# Timing experiment (MI355X box): builds the library once per set of -D switches given in VARIANTS (';'-separated, e.g.
# "-DPK_TRIP_SMALL=2;-DPK_TRIP_SMALL=3") into gpurun_out/variants/ and prints bench.py's decode-step time for each.
# The product library is never built with such switches.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/variants
IFS=';' read -ra VARS <<< "${VARIANTS:-}"
i=0
for V in "${VARS[@]}"; do
  D=$ROOT/gpurun_out/variants/v$i; mkdir -p $D; OBJS=""
  for f in $ROOT/semi_tts_amd/csrc/*.hip; do
    o=$D/$(basename $f .hip).o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=16 $V -c $f -o $o &
    OBJS="$OBJS $o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib.so $OBJS || exit 1
  for rep in 1 2; do
  ST_LIB_PATH=$D/lib.so timeout 300 python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS} 2> /dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('%-40s us_per_step %.2f' % ('$V', r['us_per_decode_step']))"
  done
  rm -rf $D
  i=$((i+1))
done
