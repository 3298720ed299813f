#!/usr/bin/env bash
# Round 4: does the compositing forward pay for a partial last round of workgroups?  Same density, tile rows 40..44.
set -o pipefail
for rows in 36 40 41 42 43 44 48; do
  H=$((rows * 16)); N=$((1000000 * H / 680))
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-tracker --no-variants --gaussians $N --height $H 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
s=d['roofline']['stage_ms']; t=75*$rows
print('rows $rows tiles', t, 'WG rounds at 1536 slots', round(t/1536,2), 'fwd us', round(s['raster_fwd']*1e3,1), 'per 1000 tiles', round(s['raster_fwd']*1e6/t,1), 'bwd us', round(s['raster_bwd']*1e3,1), 'per 1000 tiles', round(s['raster_bwd']*1e6/t,1))
"
done
