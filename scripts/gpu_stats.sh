#!/usr/bin/env bash
# Dev: trip statistics of the compositing backward (build_ab/lib_stats.so, -DGSL_G16_STATS) at R.
set -o pipefail
cp gsplatloc_amd/libgsloc_hip.so build_ab/lib_default.so
cp build_ab/lib_stats.so gsplatloc_amd/libgsloc_hip.so
timeout -k 10 300 python3 scripts/g16_stats.py 1.0; rc=$?
cp build_ab/lib_default.so gsplatloc_amd/libgsloc_hip.so
exit $rc
