#!/usr/bin/env bash
# Dev: the drop-in API with the cached context: tests that go through gsplat.rasterization, wall time, tracker rates.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_tracker.py tests/test_gpu_eval.py -q -m gpu -x > gpurun_out/dropin_tests.log 2>&1; rc=$?
tail -6 gpurun_out/dropin_tests.log
if [ $rc -ne 0 ]; then grep -n "Error\|assert" gpurun_out/dropin_tests.log | head -20; exit $rc; fi
echo "== cached"; timeout -k 10 300 python3 scripts/api_bench.py 2>/dev/null
echo "== per-call allocation"; GSLOC_DROPIN_CACHE=0 timeout -k 10 300 python3 scripts/api_bench.py 2>/dev/null
timeout -k 10 600 python3 scripts/bench_tracker.py S graph,context,autograd 2>/dev/null | tee gpurun_out/tracker_S.json | cut -c1-900
