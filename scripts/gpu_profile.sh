#!/usr/bin/env bash
# One GPU call that produces everything the round's profiles/ and BASELINE.md table are made of.
set -o pipefail
mkdir -p gpurun_out
R=$(pwd)
guard() { if [ "$1" -ge 124 ]; then echo "step ended with rc=$1: stopping"; exit "$1"; fi; }
line() { python3 - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
st = {k: round(v, 4) for k, v in d['roofline']['stage_ms'].items()}
print(f"{d['config']['workload'][:60]:60s} {d['ms_per_step']:.4f} ms {d['value']:.4g} G/s events {d['step_ms_hip_events']} stages {st} frac {d['roofline']['frac']:.3f}", flush=True)
if d.get('parity'): print('   parity', json.dumps(d['parity']))
if d.get('cpu_baseline'): print('   cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
if d.get('pose_opt'): print('   pose_opt', d['pose_opt'])
PY
}
timeout -k 10 600 python bench.py > gpurun_out/bench_R.json 2> gpurun_out/bench_R.err; rc=$?; guard $rc; [ $rc -eq 0 ] && line gpurun_out/bench_R.json || tail -5 gpurun_out/bench_R.err
for wl in S T X; do
  timeout -k 10 600 python bench.py --workload $wl --no-tracker --no-variants > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err; rc=$?; guard $rc
  [ $rc -eq 0 ] && line gpurun_out/bench_$wl.json || tail -5 gpurun_out/bench_$wl.err
done
timeout -k 10 600 python bench.py --workload X --staging fp32 --no-tracker --no-variants --no-cpu-baseline > gpurun_out/bench_X32.json 2> gpurun_out/bench_X32.err; rc=$?; guard $rc
[ $rc -eq 0 ] && line gpurun_out/bench_X32.json || tail -5 gpurun_out/bench_X32.err
cd /tmp && export TMPDIR=/tmp && cd "$R"
B="--no-cpu-baseline --no-tracker --no-variants"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_R -o st --output-format csv -- python3 bench.py $B --steps 10 --warmup 3 > gpurun_out/prof_R.log 2>&1; guard $?
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_X -o st --output-format csv -- python3 bench.py $B --workload X --steps 10 --warmup 3 > gpurun_out/prof_X.log 2>&1; guard $?
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py $B --no-graph --steps 3 --warmup 1 > gpurun_out/pmc_fetch.log 2>&1; guard $?
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py $B --no-graph --steps 3 --warmup 1 > gpurun_out/pmc_write.log 2>&1; guard $?
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_gather16 -o g --output-format csv -- scripts/ubench/gather16.bin > gpurun_out/gather16.log 2>&1; guard $?
cat gpurun_out/gather16.log | tail -4
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/pmc_gather16/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE": print(r["Kernel_Name"][:30], "FETCH_SIZE KiB", r["Counter_Value"])
PY
find gpurun_out/prof_R gpurun_out/prof_X -name "*kernel_stats.csv" | head
