#!/usr/bin/env bash
# Dev: long tile lists: pile timing with and without the split, parity tests on the pile and T frames, guards.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python3 scripts/pile_bench.py 2>&1 | tail -1
GSLOC_LONG_LISTS=0 timeout -k 10 200 python3 scripts/pile_bench.py 2>&1 | tail -1
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_guards.py tests/test_gpu_tracker.py -q -x -k "config_T or long or pile or overflow or recover" > gpurun_out/long_tests.log 2>&1; rc=$?
tail -3 gpurun_out/long_tests.log; grep -a "^E  " gpurun_out/long_tests.log | head
exit $rc
