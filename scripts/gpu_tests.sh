#!/usr/bin/env bash
# The GPU test suite the driver runs, plus smoke, in one call; logs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -s > gpurun_out/t_all.log 2>&1
rc=$?; tail -5 gpurun_out/t_all.log; grep "parity\]" gpurun_out/t_all.log | cut -c1-400
if grep -aq "Memory access fault" gpurun_out/t_all.log; then echo "GPU FAULT in the test run"; exit 1; fi
if [ "$rc" -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
