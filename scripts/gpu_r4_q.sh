#!/usr/bin/env bash
# Round 4 quick check after a kernel change: parity subset + stage times at R (and D).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guards.py tests/test_gpu_reorder.py tests/test_gpu_stress.py -q -x > gpurun_out/quick_tests.log 2>&1; rc=$?
tail -3 gpurun_out/quick_tests.log
if [ $rc -ne 0 ]; then grep -a "^E  \|^FAILED" gpurun_out/quick_tests.log | head; exit $rc; fi
pl() { python3 -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1])
print('$2', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline']['stage_ms'].items()}, d.get('step_ms_hip_events'))
"; }
B="--no-cpu-baseline --no-tracker --no-variants"
timeout -k 10 300 python bench.py $B > gpurun_out/q_R.json 2> gpurun_out/q_R.err || { tail -5 gpurun_out/q_R.err; exit 1; }
pl gpurun_out/q_R.json R
timeout -k 10 300 python bench.py $B --workload D > gpurun_out/q_D.json 2> gpurun_out/q_D.err || { tail -5 gpurun_out/q_D.err; exit 1; }
pl gpurun_out/q_D.json D
GSLOC_BWD=general timeout -k 10 300 python bench.py $B --workload D > gpurun_out/q_Dg.json 2> gpurun_out/q_Dg.err || { tail -5 gpurun_out/q_Dg.err; exit 1; }
pl gpurun_out/q_Dg.json D-general
