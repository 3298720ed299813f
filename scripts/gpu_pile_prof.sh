#!/usr/bin/env bash
# Dev: per-kernel times of the pile frame (scripts/pile_bench.py) under rocprofv3.
set -o pipefail
mkdir -p gpurun_out/pile_prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/pile_prof -o pile --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/pile_bench.py > $GRAFT_REPO_ROOT/gpurun_out/pile_prof/run.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/pile_prof/run.log; exit 1; }
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/pile_prof/run.log
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/pile_prof/**/pile_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print(r['Name'][:70].ljust(70), r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
