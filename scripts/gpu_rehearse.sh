#!/usr/bin/env bash
# Two ranks on ONE GPU (gloo through the host): the N > 1 path of bench.py with graph replay, staged so that at most
# one stage can fault (each stage is its own pair of processes; the first failure stops the script).
#   stage 1: the shipped path -- graph of library launches + pack kernel, pinned host buffer for the rehearsal
#   stage 2: + round 1's pageable host copies between replays (tensor.cpu() / copy_ from a pageable tensor)
#   stage 3: + round 1's captured torch copy node instead of the pack kernel (pinned host path)
set -o pipefail
mkdir -p gpurun_out
export GSLOC_BENCH_TRACE=1
B="bench.py --gpus 2 --rehearse-on-one-gpu --steps 12 --warmup 6 --no-cpu-baseline --no-tracker --no-variants"
run() {
  local tag=$1; shift
  timeout -k 10 240 env "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29611 $B > "gpurun_out/rehearse_${tag}.log" 2>&1
  local rc=$?
  grep -E "Memory access fault|^\{" "gpurun_out/rehearse_${tag}.log" | cut -c1-260
  echo "stage ${tag}: rc=${rc}"
  if [ $rc -ne 0 ]; then tail -5 "gpurun_out/rehearse_${tag}.log"; exit $rc; fi
}
run 1_shipped GSLOC_X=0
run 2_pageable GSLOC_DIAG_PAGEABLE=1
run 3_captured_copy GSLOC_DIAG_CAPTURED_COPY=1
run 4_both GSLOC_DIAG_CAPTURED_COPY=1 GSLOC_DIAG_PAGEABLE=1
