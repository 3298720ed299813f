#!/usr/bin/env bash
# Dev: A/B of the long-list segment size (build_ab/lib_seg<N>.so built with make EXTRA=-DGSL_SEG_LOG2=..): pile tests +
# per-kernel times of the pile frame.  usage: gpu_seg_ab.sh 128 64
set -o pipefail
mkdir -p gpurun_out
cp gsplatloc_amd/libgsloc_hip.so build_ab/lib_default.so
for v in "$@"; do
  echo "== seg $v"
  cp build_ab/lib_seg$v.so gsplatloc_amd/libgsloc_hip.so
  timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_guards.py tests/test_gpu_tracker.py -q -x -k "pile or long" > gpurun_out/seg_$v.log 2>&1 || { tail -20 gpurun_out/seg_$v.log; cp build_ab/lib_default.so gsplatloc_amd/libgsloc_hip.so; exit 1; }
  tail -1 gpurun_out/seg_$v.log
  bash scripts/gpu_pile_prof.sh || exit 1
done
cp build_ab/lib_default.so gsplatloc_amd/libgsloc_hip.so
