#!/usr/bin/env bash
# Dev tool: kernel durations of the tracker iteration on a 640x480 depth frame with every pixel a Gaussian (307 k).
set -o pipefail
mkdir -p gpurun_out
R=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$R"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_T -o st --output-format csv -- python3 scripts/bench_tracker.py T graph > gpurun_out/prof_T.log 2>&1
rc=$?; if [ $rc -ge 124 ]; then exit $rc; fi
f=$(find gpurun_out/prof_T -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.5: print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us  {r["Percentage"]}%')
PY
