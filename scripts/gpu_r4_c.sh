#!/usr/bin/env bash
# Round 4: whole GPU suite + the workgroup sort at R.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/t_all.log 2>&1
rc=$?; tail -5 gpurun_out/t_all.log
if [ $rc -ne 0 ]; then grep -a "^E  " gpurun_out/t_all.log | head -20; fi
if grep -aq "Memory access fault" gpurun_out/t_all.log; then echo "GPU FAULT in the test run"; exit 1; fi
exit $rc
