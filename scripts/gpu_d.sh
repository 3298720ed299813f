#!/usr/bin/env bash
# Dev: workload D / sigma 0 with the tiny backward vs the general (G16) backward
for m in auto general; do
  echo "== GSLOC_BWD=$m"
  GSLOC_BWD=$m timeout -k 10 300 python3 bench.py --workload D --no-cpu-baseline --no-tracker --no-variants --steps 50 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'], 4), d['step_ms_hip_events'], {k: round(v, 4) for k, v in d['roofline']['stage_ms'].items()}, d['config']['backward'])"
  GSLOC_BWD=$m timeout -k 10 300 python3 scripts/diag_sigma0.py --sigmas 0.0 --orders random,raster --steps 100 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['order'], 'tiny', d['tiny'], 'graph median', round(d['graph']['median'], 4), 'stages', {k: round(v, 4) for k, v in d['stages'].items()})
"
done
