#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out
pl() { python3 -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1])
print('$2', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['stage_ms'].items()}, d.get('step_ms_hip_events'))
"; }
B="--no-cpu-baseline --no-tracker --no-variants"
timeout -k 10 300 python bench.py $B --workload X > gpurun_out/x1.json 2> gpurun_out/x1.err || { tail -5 gpurun_out/x1.err; exit 1; }
pl gpurun_out/x1.json X-tile-order
GSLOC_REORDER=0 timeout -k 10 300 python bench.py $B --workload X > gpurun_out/x0.json 2> gpurun_out/x0.err || { tail -5 gpurun_out/x0.err; exit 1; }
pl gpurun_out/x0.json X-as-given
timeout -k 10 300 python bench.py $B --gaussians 2000000 > gpurun_out/y1.json 2> gpurun_out/y1.err || { tail -5 gpurun_out/y1.err; exit 1; }
pl gpurun_out/y1.json 2M-tile-order
GSLOC_REORDER=0 timeout -k 10 300 python bench.py $B --gaussians 2000000 > gpurun_out/y0.json 2> gpurun_out/y0.err || { tail -5 gpurun_out/y0.err; exit 1; }
pl gpurun_out/y0.json 2M-as-given
