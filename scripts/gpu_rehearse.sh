#!/usr/bin/env bash
# Diagnosis of the "write access to a read-only page" fault of graph replay on the N > 1 path (two ranks on ONE
# GPU, gloo through the host).  Staged from the least to the most complete reproduction; every stage is its own
# process (pair); the first failure stops the script, so one call can fault at most once.
#   A: ONE process, strip mode, graph replay + pinned D2H/H2D copies between replays (no second process, no gloo)
#   B: two processes, graph replays only, dist.barrier() between them (no copies)
#   C: two processes, the shipped path (graph + pinned copies + gloo all-reduce), HIP API log kept
set -o pipefail
mkdir -p gpurun_out
export GSLOC_BENCH_TRACE=1
B="--steps 12 --warmup 6 --no-cpu-baseline --no-tracker --no-variants"
finish() {
  local tag=$1 rc=$2
  grep -E "Memory access fault|^\{" "gpurun_out/rehearse_${tag}.log" | cut -c1-200
  echo "stage ${tag}: rc=${rc}"
  if [ $rc -ne 0 ]; then grep -E "diag rank|trace rank" "gpurun_out/rehearse_${tag}.log" | tail -60; exit $rc; fi
}
GSLOC_DIAG=strip,hostcopy timeout -k 10 240 python bench.py $B > gpurun_out/rehearse_A.log 2>&1; finish A $?
GSLOC_DIAG=nocopy timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29611 bench.py --gpus 2 --rehearse-on-one-gpu $B > gpurun_out/rehearse_B.log 2>&1; finish B $?
GSLOC_DIAG=full AMD_LOG_LEVEL=3 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29612 bench.py --gpus 2 --rehearse-on-one-gpu $B > gpurun_out/rehearse_C.log 2>&1
rc=$?; tail -c 3000000 gpurun_out/rehearse_C.log > gpurun_out/rehearse_C_tail.log; rm -f gpurun_out/rehearse_C.log
grep -E "Memory access fault|^\{" gpurun_out/rehearse_C_tail.log | cut -c1-200; echo "stage C: rc=$rc"; exit $rc
