#!/usr/bin/env bash
# Dev: long tile lists split over workgroups: pile timing, parity of the T frame with a 24k-entry tile, guards.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python3 scripts/pile_bench.py 2>&1 | tail -3
GSLOC_LONG_LISTS=0 timeout -k 10 200 python3 scripts/pile_bench.py 2>&1 | tail -1
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -q -x -k "config_T or tracker_loss" > gpurun_out/long_tests.log 2>&1; rc=$?
tail -5 gpurun_out/long_tests.log; grep -a "parity\]" gpurun_out/long_tests.log | cut -c1-500
exit $rc
