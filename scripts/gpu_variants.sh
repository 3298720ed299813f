#!/usr/bin/env bash
# Dev: build the library with different macro sets and time workload R's stages.  usage: gpu_variants.sh "<flags1>" "<flags2>" ...
for fl in "$@"; do
  echo "== $fl"
  touch gsplatloc_amd/csrc/raster_g16.hip gsplatloc_amd/csrc/raster_px.hip
  make -C gsplatloc_amd/csrc EXTRA="$fl" 2>&1 | grep -E "error" -A5
  timeout -k 10 300 python3 scripts/diag_sigma0.py --sigmas 1.0 --orders random --steps 100 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['order'], 'graph median', round(d['graph']['median'], 4), 'stages', {k: round(v, 4) for k, v in d['stages'].items()})
"
done
