#!/usr/bin/env bash
# Round-end evidence in one call: the GPU test suite + smoke (parity report), then the profile run.
set -o pipefail
mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
bash scripts/gpu_tests.sh; rc=$?
grep -a "passed\|failed" gpurun_out/t_all.log | tail -2
if [ $rc -ge 124 ]; then exit $rc; fi
bash scripts/gpu_profile.sh
