#!/usr/bin/env bash
# One GPU call that produces everything the round's profiles/ and BASELINE.md table are made of (round 3).
set -o pipefail
mkdir -p gpurun_out
R=$(pwd)
guard() { if [ "$1" -ge 124 ]; then echo "step ended with rc=$1: stopping"; exit "$1"; fi; }
line() { python3 - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
st = {k: round(v, 4) for k, v in d['roofline']['stage_ms'].items()}
print(f"{d['config']['workload'][:60]:60s} {d['ms_per_step']:.4f} ms {d['value']:.4g} G/s events {d['step_ms_hip_events']} stages {st} frac {d['roofline']['frac']:.3f} whole {d['roofline']['whole_step']['frac']:.3f}", flush=True)
if d.get('parity'): print('   parity', json.dumps(d['parity'])[:600])
if d.get('cpu_baseline'): print('   cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
if d.get('pose_opt'): print('   pose_opt', d['pose_opt'])
for v in d.get('variants') or []: print('   variant', {k: v.get(k) for k in ('workload', 'sigma_px', 'order', 'ms_per_step', 'ms_per_step_wall_mean')})
print('   host', d.get('host'))
PY
}
timeout -k 10 600 python bench.py > gpurun_out/bench_R.json 2> gpurun_out/bench_R.err; rc=$?; guard $rc; [ $rc -eq 0 ] && line gpurun_out/bench_R.json || tail -5 gpurun_out/bench_R.err
for wl in D S T X; do
  timeout -k 10 600 python bench.py --workload $wl --no-tracker --no-variants > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err; rc=$?; guard $rc
  [ $rc -eq 0 ] && line gpurun_out/bench_$wl.json || tail -5 gpurun_out/bench_$wl.err
done
timeout -k 10 600 python bench.py --workload X --staging fp32 --no-tracker --no-variants --no-cpu-baseline > gpurun_out/bench_X32.json 2> gpurun_out/bench_X32.err; rc=$?; guard $rc
[ $rc -eq 0 ] && line gpurun_out/bench_X32.json || tail -5 gpurun_out/bench_X32.err
cd /tmp && export TMPDIR=/tmp && cd "$R"
B="--no-cpu-baseline --no-tracker --no-variants"
for wl in R D X; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$wl -o st --output-format csv -- python3 bench.py $B --workload $wl --steps 10 --warmup 3 > gpurun_out/prof_$wl.log 2>&1; guard $?
done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py $B --no-graph --steps 3 --warmup 1 > gpurun_out/pmc_fetch.log 2>&1; guard $?
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py $B --no-graph --steps 3 --warmup 1 > gpurun_out/pmc_write.log 2>&1; guard $?
python3 scripts/pmc_summary.py $(find gpurun_out/pmc_fetch -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_write -name "*counter_collection.csv" | head -1) gpurun_out/pmc_traffic.json
find gpurun_out/prof_R gpurun_out/prof_D gpurun_out/prof_X -name "*kernel_stats.csv" | head
