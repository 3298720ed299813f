#!/usr/bin/env bash
# One GPU call: parity of the paths that have not run on hardware yet, then the headline workload and the
# sigma->0 / raster variant under each build switch, then the timing ablations of the compositing backward.
# An ordinary failure (a test assertion, exit code 1) lets the next step run; a timeout, abort or signal
# (a hung or faulting kernel) stops the whole script -- nothing touches the GPU after that.
#
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash scripts/run_prepared_experiments.sh'
set -o pipefail
mkdir -p gpurun_out
B="--no-cpu-baseline --no-tracker --no-variants --steps 20 --warmup 3"
guard() {  # rc: stop everything on timeout / abort / kill / segv
  local rc=$1
  if [ "$rc" -ge 124 ]; then echo "step ended with rc=$rc: stopping"; exit "$rc"; fi
}
run_bench() {  # tag, extra bench args...; environment comes from the caller
  local tag=$1; shift
  timeout -k 10 240 python bench.py $B "$@" > "gpurun_out/exp_${tag}.json" 2> "gpurun_out/exp_${tag}.err"
  local rc=$?
  guard $rc
  if [ $rc -ne 0 ]; then echo "$tag: bench failed rc=$rc"; tail -5 "gpurun_out/exp_${tag}.err"; return 0; fi
  python - "$tag" "gpurun_out/exp_${tag}.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
st = {k: round(v, 4) for k, v in d['roofline']['stage_ms'].items()}
print(f"{sys.argv[1]:28s} {d['ms_per_step']:.4f} ms  {d['value']:.4g} G/s  stages {st}", flush=True)
PY
}
run_tests() {  # log name, timeout, pytest args...
  local log=$1 to=$2; shift 2
  timeout -k 10 "$to" python -m pytest "$@" -m gpu -q > "gpurun_out/${log}.log" 2>&1
  local rc=$?
  tail -4 "gpurun_out/${log}.log"
  guard $rc
}
run_bench default_s1
GSLOC_LIB_VARIANT=abl1 run_bench abl1_plain_global_s1
GSLOC_LIB_VARIANT=abl2 run_bench abl2_plain_lds_s1
GSLOC_LIB_VARIANT=abl3 run_bench abl3_no_reduce_s1
GSLOC_AOS=1 run_bench aos_s1
GSLOC_LIB_VARIANT=occ5 run_bench occ5_s1
GSLOC_LIB_VARIANT=xcd run_bench xcd_s1
run_bench default_s0_raster --sigma-px 0 --order raster
GSLOC_TINY_GATHER=4 run_bench gather4_s0_raster --sigma-px 0 --order raster
GSLOC_TINY_FUSED=1 run_bench fusedgather_s0_raster --sigma-px 0 --order raster
GSLOC_LIB_VARIANT=xcd run_bench xcd_s0_raster --sigma-px 0 --order raster
GSLOC_AOS=1 run_bench aos_s0_random --sigma-px 0 --order random
run_bench default_s0_random --sigma-px 0 --order random
GSLOC_EXPERIMENTAL=1 run_tests exp_tests 600 tests/test_gpu_experimental.py
GSLOC_FULLSIZE=1 run_tests exp_fullsize 500 tests/test_gpu_fullsize.py
