#!/usr/bin/env bash
# SQ counters of the compositing kernels (one rocprofv3 --pmc pass, eager launches).
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-tracker --no-variants --no-graph --steps 3 --warmup 1"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d gpurun_out/pmc_sq -o sq --output-format csv -- python3 bench.py $B > gpurun_out/pmc_sq.log 2>&1
rc=$?; echo "pmc rc=$rc"; if [ "$rc" -ge 124 ]; then exit $rc; fi
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0))[:9]:
    n = cnt[k]
    print(f"{k:40s} n={n:3d} " + " ".join(f"{c[3:]}={v / n:.3g}" for c, v in sorted(agg[k].items())))
PY
