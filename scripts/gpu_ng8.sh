#!/usr/bin/env bash
# Dev: the backward with 8 half-block groups (build_ab/lib_ng8.so, -DGSL_NG=8) against the built library: parity subset
# with the variant in place, then stage times of both.
set -o pipefail
mkdir -p gpurun_out
cp gsplatloc_amd/libgsloc_hip.so build_ab/lib_default.so
cp build_ab/lib_ng8.so gsplatloc_amd/libgsloc_hip.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reorder.py tests/test_gpu_stress.py -q -x > gpurun_out/ng8_tests.log 2>&1; rc=$?
tail -3 gpurun_out/ng8_tests.log
cp build_ab/lib_default.so gsplatloc_amd/libgsloc_hip.so
if [ $rc -ne 0 ]; then grep -a "^E  \|^FAILED" gpurun_out/ng8_tests.log | head; exit $rc; fi
WLS="R D X" bash scripts/gpu_lib_ab.sh default ng8
