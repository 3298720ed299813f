#!/usr/bin/env bash
# Dev: run one test with each of build_ab/lib_bis_<sha>.so in place of the built library.
# usage: gpu_bisect.sh "<pytest node ids>" <sha> [...]
set -o pipefail
tests=$1; shift
cp gsplatloc_amd/libgsloc_hip.so build_ab/lib_default.so
for v in default "$@"; do
  [ "$v" = default ] || cp build_ab/lib_bis_$v.so gsplatloc_amd/libgsloc_hip.so
  echo "== $v"
  timeout -k 10 300 python -m pytest $tests -q -s 2>&1 | grep -a "parity\]\|passed\|failed\|^E  " | cut -c1-200
done
cp build_ab/lib_default.so gsplatloc_amd/libgsloc_hip.so
