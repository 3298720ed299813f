#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_eval.py tests/test_gpu_configs.py tests/test_gpu_tracker.py tests/test_gpu_parallel.py -m gpu -q -s > gpurun_out/t_new.log 2>&1
rc=$?; tail -8 gpurun_out/t_new.log; grep -E "parity\]|\[eval\]|^frame " gpurun_out/t_new.log | cut -c1-420; exit $rc
