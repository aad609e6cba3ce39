#!/bin/bash
mkdir -p gpurun_out
timeout 300 tools/mb/mb_wstat > gpurun_out/l_wstat.txt 2>&1; cat gpurun_out/l_wstat.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "dp or rccl or train or grad or cbhg" > gpurun_out/l_pytest.log 2>&1; echo "pytest exit $?"; tail -n 6 gpurun_out/l_pytest.log
timeout 900 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/l_train.json 2> gpurun_out/l_train.err; python -c "
import json; r=json.load(open('gpurun_out/l_train.json')); print('train plain', r['ms_per_step'], r['collectives_per_step'])"
timeout 900 python bench.py --workload train --dist --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/l_train_dist.json 2> gpurun_out/l_train_dist.err; python -c "
import json; r=json.load(open('gpurun_out/l_train_dist.json')); print('train --dist', r['ms_per_step'], r['ms_variants'], r['collectives_per_step'], r['rccl_ranks'])"; tail -3 gpurun_out/l_train_dist.err
