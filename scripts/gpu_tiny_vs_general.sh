#!/usr/bin/env bash
# Dev: tiny-splat backward (auto) against the general per-quadrant backward on the sub-pixel workloads.
for m in auto general; do
  echo "== GSLOC_BWD=$m"
  for wl in S T D; do
    GSLOC_BWD=$m timeout -k 10 300 python3 bench.py --workload $wl --no-cpu-baseline --no-tracker --no-variants --steps 50 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$wl', round(d['ms_per_step'], 4), d['step_ms_hip_events']['median'], {k: round(v, 4) for k, v in d['roofline']['stage_ms'].items()})"
  done
  for c in S T R; do GSLOC_BWD=$m timeout -k 10 400 python3 scripts/bench_tracker.py $c graph 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('tracker $c', round(d['graph']['iters_per_s']), 'it/s', round(d['graph']['ms_per_iter'], 4), 'ms')"; done
  GSLOC_BWD=$m timeout -k 10 300 python3 scripts/diag_sigma0.py --sigmas 0.0 --orders random,raster --steps 100 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('sigma0', d['order'], 'tiny', d['tiny'], 'graph median', round(d['graph']['median'], 4))
"
done
