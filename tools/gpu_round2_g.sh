#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grad.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "gru or whole_forward or golden or postnet or c1" 2>&1 | tail -5
timeout 300 python tools/bench_full_forward.py 2>/dev/null | tail -1
