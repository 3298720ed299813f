#!/usr/bin/env bash
# Dev: rocprofv3 kernel statistics of the library's kernels at R (10 graph replays).
set -o pipefail
mkdir -p gpurun_out
R=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$R"
rm -rf gpurun_out/kst
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kst -o st --output-format csv -- python3 bench.py --no-cpu-baseline --no-tracker --no-variants --steps 10 --warmup 3 ${WL:+--workload $WL} > gpurun_out/kst.log 2>&1 || { tail -5 gpurun_out/kst.log; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kst/**/st_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "gsl::" in r["Name"]:
        print("%-44s calls %4s avg %8.1f us" % (r["Name"].replace("void gsl::", "")[:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
