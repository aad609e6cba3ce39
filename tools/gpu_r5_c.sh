#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_grad.py -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -3
for V in "ST_DXD_SPLITS=2"; do
  echo "== $V"
  env $V timeout 600 python bench.py --workload train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['last'])"
  env $V bash tools/gpu_train_prof.sh r05x | grep -E "steady|pk_pw_ab|pk_pw_hist|pk_part|pk_kernel<1, 1"
done
