#!/usr/bin/env bash
# Diagnosis of the "write access to a read-only page" fault: ONE process, graph replay with something else issued on
# the stream between two replays (stage A of the previous script reproduced it: strip + pinned D2H / H2D copies).
# Every stage is its own process; the first failure stops the script, so one call faults at most once.
set -o pipefail
mkdir -p gpurun_out
export GSLOC_BENCH_TRACE=1
B="--steps 12 --warmup 6 --no-cpu-baseline --no-tracker --no-variants"
stage() {
  local tag=$1; shift
  timeout -k 10 240 env "$@" python bench.py $B > "gpurun_out/rehearse_${tag}.log" 2>&1
  local rc=$?
  grep -E "Memory access fault" "gpurun_out/rehearse_${tag}.log" | cut -c1-200
  echo "stage ${tag}: rc=${rc}"
  if [ $rc -ne 0 ]; then tail -c 400000 "gpurun_out/rehearse_${tag}.log" > "gpurun_out/rehearse_${tag}_tail.log"; rm -f "gpurun_out/rehearse_${tag}.log"; exit $rc; fi
}
stage S1_eager_kernel_between_replays GSLOC_DIAG=strip,eagerkernel
stage S2_copies_on_side_stream GSLOC_DIAG=strip,hostcopy,sidestream
stage S3_host_kernarg GSLOC_DIAG=strip,hostcopy HIP_FORCE_DEV_KERNARG=0
stage S4_d2h_only GSLOC_DIAG=strip,hostcopy,d2honly
stage S5_h2d_only GSLOC_DIAG=strip,hostcopy,h2donly
stage S6_full_frame_copies_logged GSLOC_DIAG=hostcopy AMD_LOG_LEVEL=3
