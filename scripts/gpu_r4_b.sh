#!/usr/bin/env bash
# Round 4 dev call: stage times at R for the three placements (as given / tile order / raster-ordered input).
set -o pipefail
mkdir -p gpurun_out
pl() { python3 -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1])
print('$2', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['stage_ms'].items()}, d.get('step_ms_hip_events'))
"; }
B="--no-cpu-baseline --no-tracker --no-variants"
timeout -k 10 300 python bench.py $B > gpurun_out/b_tile.json 2> gpurun_out/b_tile.err || { tail -5 gpurun_out/b_tile.err; exit 1; }
pl gpurun_out/b_tile.json tile-order
GSLOC_REORDER=0 timeout -k 10 300 python bench.py $B --order raster > gpurun_out/b_raster.json 2> gpurun_out/b_raster.err || { tail -5 gpurun_out/b_raster.err; exit 1; }
pl gpurun_out/b_raster.json raster-input
timeout -k 10 300 python bench.py $B --order raster > gpurun_out/b_raster2.json 2> gpurun_out/b_raster2.err || { tail -5 gpurun_out/b_raster2.err; exit 1; }
pl gpurun_out/b_raster2.json raster-input-auto
