#!/bin/bash
# Timing experiment (MI355X box): what does each launch of the decode step cost IN SITU (un-profiled hipGraph replay)?
# Builds a separate library with -DST_ABLATE (the product library has no such switch) and times bench.py with one launch of the
# step skipped at a time (outputs are garbage in those runs; only the step time is read).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/ablate
OBJS=""
for f in $ROOT/semi_tts_amd/csrc/*.hip; do
  o=$ROOT/gpurun_out/ablate/$(basename $f .hip).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=16 -DST_ABLATE -c $f -o $o &
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/gpurun_out/ablate/libsemitts_ablate.so $OBJS || exit 1
export ST_LIB_PATH=$ROOT/gpurun_out/ablate/libsemitts_ablate.so
for EXTRA in "${ABL_ARGS:-}"; do
for SKIP in ${SKIPS:-0 1 6 8 16 32 64 128 192 63}; do      # (2 | 4 = the merged query-projection + attention-fin launch)
  ST_SKIP=$SKIP timeout 300 python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-finite-check $EXTRA 2> /dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('skip=%-3s %-9s us_per_step %.2f' % ('$SKIP', '$EXTRA', r['us_per_decode_step']))"
done
done
rm -f $ROOT/gpurun_out/ablate/*.o $ROOT/gpurun_out/ablate/*.so
