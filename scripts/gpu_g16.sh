#!/usr/bin/env bash
# Dev: parity subset + A/B timing of the G16 backward against the quadrant/MFMA backward.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guards.py tests/test_gpu_stress.py -q > gpurun_out/g16_tests.log 2>&1; rc=$?
tail -15 gpurun_out/g16_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for k in g16 mfma; do
  echo "== $k"
  GSLOC_BWD_KERNEL=$k timeout -k 10 300 python3 scripts/diag_sigma0.py --sigmas 1.0 --orders random,raster --steps 100 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['order'], 'graph median', round(d['graph']['median'], 4), 'stages', {k: round(v, 4) for k, v in d['stages'].items()})
    else: print(l.rstrip()[:300])
"
done
