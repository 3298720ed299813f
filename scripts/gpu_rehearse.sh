#!/usr/bin/env bash
# Rehearsal of the N > 1 path of bench.py on ONE GPU: two ranks, each replays its HIP graph per step, the 16-float
# all-reduce goes through gloo and a pinned host buffer.  (Round 1's "write access to a read-only page" fault of this
# path came from the hipMemsetAsync node inside the captured iteration)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 \
  bench.py --gpus 2 --rehearse-on-one-gpu --steps 20 --warmup 5 > gpurun_out/rehearse_2ranks.log 2>&1
rc=$?; grep -E "Memory access fault|^\{" gpurun_out/rehearse_2ranks.log | cut -c1-400; echo "two ranks: rc=$rc"; exit $rc
