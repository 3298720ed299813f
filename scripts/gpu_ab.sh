#!/usr/bin/env bash
# Dev: A/B timing of the backward kernels at R (sigma 1) -- usage: gpu_ab.sh [kernels...]
set -o pipefail
for k in "${@:-g16 mfma}"; do
  echo "== $k"
  GSLOC_BWD_KERNEL=$k timeout -k 10 300 python3 scripts/diag_sigma0.py --sigmas 1.0 --orders random --steps 100 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['order'], 'graph median', round(d['graph']['median'], 4), 'stages', {k: round(v, 4) for k, v in d['stages'].items()})
"
done
