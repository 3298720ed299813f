#!/usr/bin/env bash
# One GPU call: sigma->0 R-size diagnosis (per-step event distribution + rocprofv3 kernel trace of the bench command).
set -o pipefail
mkdir -p gpurun_out
R=$(pwd)
timeout -k 10 400 python3 scripts/diag_sigma0.py --sigmas 0.0,1.0 > gpurun_out/diag_sigma0.log 2> gpurun_out/diag_sigma0.err; rc=$?
cat gpurun_out/diag_sigma0.log; tail -3 gpurun_out/diag_sigma0.err
if [ $rc -ge 124 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp && cd "$R"
for o in raster random; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ks_s0_$o -o st --output-format csv -- python3 bench.py --sigma-px 0 --order $o --no-cpu-baseline --no-tracker --no-variants --steps 20 --warmup 3 > gpurun_out/ks_s0_$o.log 2>&1
rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
tail -1 gpurun_out/ks_s0_$o.log | cut -c1-600
f=$(find gpurun_out/ks_s0_$o -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.3: print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us min {float(r["MinNs"])/1e3:8.2f} max {float(r["MaxNs"])/1e3:8.2f} {r["Percentage"]}%')
PY
done
