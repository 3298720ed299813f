#!/usr/bin/env bash
# Dev: run scripts/gpu_quick.sh (parity subset + stage times at R, D, pile) with build_ab/lib_<name>.so in place of the
# built library.  usage: gpu_lib_ab.sh <name> [...]   ("default" = the library as built)
set -o pipefail
cp gsplatloc_amd/libgsloc_hip.so build_ab/lib_default.so
rc=0
for v in "$@"; do
  echo "== $v"
  cp build_ab/lib_$v.so gsplatloc_amd/libgsloc_hip.so
  bash scripts/gpu_quick.sh || { rc=1; break; }
done
cp build_ab/lib_default.so gsplatloc_amd/libgsloc_hip.so
exit $rc
