#!/bin/bash
mkdir -p gpurun_out
timeout 300 tools/mb/mb_wstat > gpurun_out/m_wstat.txt 2>&1; cat gpurun_out/m_wstat.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "dp or rccl" > gpurun_out/m_pytest.log 2>&1; echo "pytest exit $?"; tail -n 4 gpurun_out/m_pytest.log
timeout 900 python bench.py --workload train --dist --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/m_train_dist.json 2> gpurun_out/m_train_dist.err; python -c "
import json; r=json.load(open('gpurun_out/m_train_dist.json')); print('train --dist', r['ms_per_step'], r['ms_variants'], r['collectives_per_step'], r['rccl_ranks'])"; tail -3 gpurun_out/m_train_dist.err
