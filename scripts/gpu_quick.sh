#!/usr/bin/env bash
# Dev: parity subset (compositing paths, strips, guards, long lists) + stage timing at R and D.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guards.py tests/test_gpu_configs.py -q -x -k "rasterize_fwd_bwd or end_to_end or fuzz or strip or outside or fused_full or edge or long or tiny or legacy" > gpurun_out/quick_tests.log 2>&1; rc=$?
tail -3 gpurun_out/quick_tests.log
if [ $rc -ne 0 ]; then grep -a "^E  " gpurun_out/quick_tests.log | head; exit $rc; fi
timeout -k 10 300 python3 scripts/diag_sigma0.py --sigmas 1.0 --orders random --steps 100 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('R', d['order'], 'graph median', round(d['graph']['median'], 4), 'stages', {k: round(v, 4) for k, v in d['stages'].items()})
"
for m in auto general; do
GSLOC_BWD=$m timeout -k 10 300 python3 bench.py --workload D --no-cpu-baseline --no-tracker --no-variants --steps 50 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('D $m', round(d['ms_per_step'], 4), {k: round(v, 4) for k, v in d['roofline']['stage_ms'].items()})"
done
timeout -k 10 200 python3 scripts/pile_bench.py 2>&1 | tail -1
