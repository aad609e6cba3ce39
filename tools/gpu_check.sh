#!/bin/bash
# Runs on the MI355X box (via gpurun): GPU parity tests, smoke, a short bench.  Logs -> gpurun_out/
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
echo "== device" > gpurun_out/check.log
(rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8) >> gpurun_out/check.log 2>&1
echo "== pytest -m gpu" >> gpurun_out/check.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/check.log
tail -n 60 gpurun_out/pytest_gpu.log
echo "== smoke" >> gpurun_out/check.log
timeout 300 python __graft_entry__.py smoke >> gpurun_out/check.log 2>&1
echo "smoke exit $?" >> gpurun_out/check.log
if [ -z "${SKIP_BENCH}" ]; then
  echo "== bench" >> gpurun_out/check.log
  timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err
  echo "bench exit $?" >> gpurun_out/check.log
  cat gpurun_out/bench.json
  tail -n 5 gpurun_out/bench.err
fi
tail -n 20 gpurun_out/check.log
