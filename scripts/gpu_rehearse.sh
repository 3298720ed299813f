#!/usr/bin/env bash
# Rehearsal of the N > 1 path of bench.py on ONE GPU: 2 and 4 ranks, each replays its HIP graph per step, the 16-float
# all-reduce goes through gloo and a pinned host buffer.  (Round 1's "write access to a read-only page" fault of this
# path came from the hipMemsetAsync node inside the captured iteration)
set -o pipefail
mkdir -p gpurun_out
for n in 2 4; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2961$n \
    bench.py --gpus $n --rehearse-on-one-gpu --steps 20 --warmup 5 > gpurun_out/rehearse_${n}ranks.log 2>&1
  rc=$?; grep -E "Memory access fault|^\{" gpurun_out/rehearse_${n}ranks.log | cut -c1-500; echo "$n ranks: rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/rehearse_${n}ranks.log; exit $rc; fi
done
