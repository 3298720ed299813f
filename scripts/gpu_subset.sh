#!/usr/bin/env bash
# Dev tool: run a subset of the GPU tests on the box.  usage: gpu_subset.sh <log name> <pytest args...>
set -o pipefail
mkdir -p gpurun_out
log=$1; shift
timeout -k 10 1000 python -m pytest "$@" -m gpu -q -s > "gpurun_out/${log}.log" 2>&1
rc=$?; tail -12 "gpurun_out/${log}.log"; grep -E "parity\]|\[eval\]" "gpurun_out/${log}.log" | cut -c1-420; exit $rc
