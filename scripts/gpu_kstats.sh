#!/usr/bin/env bash
# Dev tool: rocprofv3 kernel statistics of bench.py for one workload.  usage: gpu_kstats.sh <workload> [extra bench args]
set -o pipefail
mkdir -p gpurun_out
R=$(pwd); wl=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$R"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ks_$wl -o st --output-format csv -- python3 bench.py --workload $wl --no-cpu-baseline --no-tracker --no-variants --steps 10 --warmup 3 "$@" > gpurun_out/ks_$wl.log 2>&1
rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
f=$(find gpurun_out/ks_$wl -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.3: print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us  {r["Percentage"]}%')
PY
