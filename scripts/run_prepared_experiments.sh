#!/usr/bin/env bash
# First GPU call of the next round: parity of the paths prepared after round 1's GPU access ended, then the
# headline workload and the sigma->0 / raster variant under each switch.  Everything is chained with && so that
# nothing runs after a failure (a faulting kernel must not be re-run), each step has its own timeout, and all
# output goes to gpurun_out/.
#
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash scripts/run_prepared_experiments.sh'
set -o pipefail
mkdir -p gpurun_out
B="--no-cpu-baseline --no-tracker --no-variants --steps 20 --warmup 3"
run_bench() {  # tag, extra bench args...; environment comes from the caller
  local tag=$1; shift
  timeout -k 10 240 python bench.py $B "$@" > "gpurun_out/exp_${tag}.json" 2> "gpurun_out/exp_${tag}.err" &&
    python - "$tag" "gpurun_out/exp_${tag}.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"{sys.argv[1]:28s} {d['ms_per_step']:.4f} ms  {d['value']:.4g} G/s  stages {d['roofline']['stage_ms']}")
PY
}
GSLOC_EXPERIMENTAL=1 timeout -k 10 600 python -m pytest tests/test_gpu_experimental.py -m gpu -q -x > gpurun_out/exp_tests.log 2>&1 &&
  tail -3 gpurun_out/exp_tests.log &&
  GSLOC_FULLSIZE=1 timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/exp_fullsize.log 2>&1 &&
  tail -3 gpurun_out/exp_fullsize.log &&
  run_bench default_s1 &&
  GSLOC_AOS=1 run_bench aos_s1 &&
  GSLOC_LIB_VARIANT=occ5 run_bench occ5_s1 &&
  GSLOC_LIB_VARIANT=xcd run_bench xcd_s1 &&
  run_bench default_s0_raster --sigma-px 0 --order raster &&
  GSLOC_TINY_GATHER=4 run_bench gather4_s0_raster --sigma-px 0 --order raster &&
  GSLOC_TINY_FUSED=1 run_bench fusedgather_s0_raster --sigma-px 0 --order raster &&
  GSLOC_LIB_VARIANT=xcd run_bench xcd_s0_raster --sigma-px 0 --order raster &&
  GSLOC_AOS=1 run_bench aos_s0_random --sigma-px 0 --order random &&
  run_bench default_s0_random --sigma-px 0 --order random
