#!/usr/bin/env bash
# Dev: long-list tests, then per-kernel times of the pile frame.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_guards.py tests/test_gpu_tracker.py -q -x -k "pile or long" > gpurun_out/pile_tests.log 2>&1 || { tail -20 gpurun_out/pile_tests.log; exit 1; }
tail -1 gpurun_out/pile_tests.log
bash scripts/gpu_pile_prof.sh
