#!/usr/bin/env bash
# A/B timing of the backward variants on one GPU (dev tool).  Ordinary failures continue, timeouts/aborts stop.
set -o pipefail
mkdir -p gpurun_out
B="--no-cpu-baseline --no-tracker --no-variants --steps 20 --warmup 3"
guard() { if [ "$1" -ge 124 ]; then echo "step ended with rc=$1: stopping"; exit "$1"; fi; }
run_bench() {
  local tag=$1; shift
  timeout -k 10 240 python bench.py $B "$@" > "gpurun_out/ab_${tag}.json" 2> "gpurun_out/ab_${tag}.err"
  local rc=$?; guard $rc
  if [ $rc -ne 0 ]; then echo "$tag: bench failed rc=$rc"; tail -5 "gpurun_out/ab_${tag}.err"; return 0; fi
  python - "$tag" "gpurun_out/ab_${tag}.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
st = {k: round(v, 4) for k, v in d['roofline']['stage_ms'].items()}
print(f"{sys.argv[1]:24s} {d['ms_per_step']:.4f} ms  {d['value']:.4g} G/s  [{d['config'].get('backward')}] {st}", flush=True)
PY
}
run_tests() {
  local log=$1 to=$2; shift 2
  timeout -k 10 "$to" python -m pytest "$@" -m gpu -q > "gpurun_out/${log}.log" 2>&1
  local rc=$?; tail -6 "gpurun_out/${log}.log"; guard $rc
}
run_tests t_mfma 600 tests/test_gpu_parity.py -k "fused_full or end_to_end or tile_strip or small_splat"
run_bench s1_mfma
GSLOC_RASTER_BWD=quad run_bench s1_quad
GSLOC_BWD=general run_bench s0r_mfma --sigma-px 0 --order raster
GSLOC_BWD=general GSLOC_RASTER_BWD=quad run_bench s0r_quad --sigma-px 0 --order raster
run_bench s1_mfma_poseonly --pose-only
run_tests t_parity 900 tests/test_gpu_parity.py
run_tests t_configs 900 tests/test_gpu_configs.py
run_tests t_rest 900 tests --ignore=tests/test_gpu_parity.py --ignore=tests/test_gpu_configs.py
