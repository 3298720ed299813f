#!/usr/bin/env bash
# Dev: stats (if built with -DGSL_G16_STATS) + quick parity (subset) + timing of the G16 backward at R.
set -o pipefail
mkdir -p gpurun_out
python3 scripts/g16_stats.py 1.0 2>/dev/null | tail -1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "rasterize_fwd_bwd or end_to_end or fuzz or strip" > gpurun_out/g16_tests.log 2>&1; rc=$?
tail -4 gpurun_out/g16_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for k in g16 mfma; do
  echo "== $k"
  GSLOC_BWD_KERNEL=$k timeout -k 10 300 python3 scripts/diag_sigma0.py --sigmas 1.0 --orders random --steps 100 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['order'], 'graph median', round(d['graph']['median'], 4), 'stages', {k: round(v, 4) for k, v in d['stages'].items()})
"
done
