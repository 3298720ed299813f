#!/usr/bin/env bash
# One GPU call that produces what the round's profiles/ are made of (round 4): bench lines for R / D / S / T / X, rocprofv3
# kernel statistics for R / D / X and the tracker at S, PMC passes (FETCH_SIZE, WRITE_SIZE, SQ instruction counts, SQ LDS
# counters) at R condensed into gpurun_out/pmc_traffic.json (copy to profiles/r04_pmc_traffic.json).
set -o pipefail
mkdir -p gpurun_out
R=$(pwd)
guard() { if [ "$1" -ge 124 ]; then echo "step ended with rc=$1: stopping"; exit "$1"; fi; }
line() { python3 - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
st = {k: round(v, 4) for k, v in d['roofline']['stage_ms'].items()}
print(f"{d['config']['workload'][:60]:60s} {d['ms_per_step']:.4f} ms {d['value']:.4g} G/s events {d['step_ms_hip_events']} stages {st} frac {d['roofline']['frac']:.3f} whole {d['roofline']['whole_step']['frac']:.3f}", flush=True)
PY
}
timeout -k 10 900 python bench.py > gpurun_out/bench_R.json 2> gpurun_out/bench_R.err; rc=$?; guard $rc; [ $rc -eq 0 ] && line gpurun_out/bench_R.json || tail -5 gpurun_out/bench_R.err
for wl in D S T X; do
  timeout -k 10 600 python bench.py --workload $wl --no-tracker --no-variants > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err; rc=$?; guard $rc
  [ $rc -eq 0 ] && line gpurun_out/bench_$wl.json || tail -5 gpurun_out/bench_$wl.err
done
cd /tmp && export TMPDIR=/tmp && cd "$R"
B="--no-cpu-baseline --no-tracker --no-variants"
for wl in R D X; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$wl -o st --output-format csv -- python3 bench.py $B --workload $wl --steps 10 --warmup 3 > gpurun_out/prof_$wl.log 2>&1; guard $?
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_trk -o st --output-format csv -- python3 scripts/bench_tracker.py S graph > gpurun_out/prof_trk.log 2>&1; guard $?
P="--no-graph --steps 3 --warmup 1"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py $B $P > gpurun_out/pmc_fetch.log 2>&1; guard $?
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py $B $P > gpurun_out/pmc_write.log 2>&1; guard $?
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d gpurun_out/pmc_sq -o sq --output-format csv -- python3 bench.py $B $P > gpurun_out/pmc_sq.log 2>&1; guard $?
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_BUSY_CU_CYCLES --kernel-trace -d gpurun_out/pmc_lds -o lds --output-format csv -- python3 bench.py $B $P > gpurun_out/pmc_lds.log 2>&1; guard $?
f() { find gpurun_out/$1 -name "*counter_collection.csv" | head -1; }
python3 scripts/pmc_summary.py $(f pmc_fetch) $(f pmc_write) gpurun_out/pmc_traffic.json $(f pmc_sq) $(f pmc_lds) | head -12
find gpurun_out/prof_R gpurun_out/prof_D gpurun_out/prof_X gpurun_out/prof_trk -name "*kernel_stats.csv" | head
