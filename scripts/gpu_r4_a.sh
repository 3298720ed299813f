#!/usr/bin/env bash
# Round 4 dev call: reorder tests + bench R with / without the tile-order placement + stage times.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_reorder.py -x -q > gpurun_out/t_reorder.log 2>&1; rc=$?
tail -5 gpurun_out/t_reorder.log
if [ $rc -ne 0 ]; then grep -a "^E  " gpurun_out/t_reorder.log | head -20; exit $rc; fi
pl() { python3 -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1])
print('$2', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['stage_ms'].items()}, d.get('step_ms_hip_events'))
p=d.get('parity')
if p: print('   parity', json.dumps(p)[:500])
"; }
timeout -k 10 300 python bench.py --no-cpu-baseline --no-tracker --no-variants > gpurun_out/r04_bench_R_reorder.json 2> gpurun_out/r04_bench_R_reorder.err || { tail -5 gpurun_out/r04_bench_R_reorder.err; exit 1; }
pl gpurun_out/r04_bench_R_reorder.json reorder
GSLOC_REORDER=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-tracker --no-variants > gpurun_out/r04_bench_R_noreorder.json 2> gpurun_out/r04_bench_R_noreorder.err || { tail -5 gpurun_out/r04_bench_R_noreorder.err; exit 1; }
pl gpurun_out/r04_bench_R_noreorder.json as-given
