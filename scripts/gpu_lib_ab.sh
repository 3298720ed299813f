#!/usr/bin/env bash
# Dev: stage times at R / D with build_ab/lib_<name>.so in place of the built library ("default" = as built).
# usage: gpu_lib_ab.sh <name> [...]
set -o pipefail
cp gsplatloc_amd/libgsloc_hip.so build_ab/lib_default.so
for v in "$@"; do
  cp build_ab/lib_$v.so gsplatloc_amd/libgsloc_hip.so
  for wl in ${WLS:-R D}; do
    timeout -k 10 300 python3 bench.py --workload $wl --no-cpu-baseline --no-tracker --no-variants --steps 40 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$v $wl', round(d['ms_per_step'], 4), {k: round(v, 4) for k, v in d['roofline']['stage_ms'].items()})"
  done
done
cp build_ab/lib_default.so gsplatloc_amd/libgsloc_hip.so
