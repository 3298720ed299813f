#!/usr/bin/env bash
# Dev: tracker-side quick check: tracker / config / multirank tests, tracker iteration rates, D / S / T stage times.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_tracker.py tests/test_gpu_configs.py tests/test_gpu_eval.py -q -x > gpurun_out/trk_tests.log 2>&1; rc=$?
tail -3 gpurun_out/trk_tests.log
if [ $rc -ne 0 ]; then grep -a "^E  \|^FAILED" gpurun_out/trk_tests.log | head; exit $rc; fi
timeout -k 10 300 python scripts/trk_render_mode.py 2>/dev/null
WLS="D S T" bash scripts/gpu_lib_ab.sh default
