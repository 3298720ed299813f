#!/usr/bin/env bash
# LDS-side SQ counters of the compositing kernels at R (two rocprofv3 --pmc passes, eager launches).
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-tracker --no-variants --no-graph --steps 3 --warmup 1"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d gpurun_out/pmc_lds -o lds --output-format csv -- python3 bench.py $B > gpurun_out/pmc_lds.log 2>&1
rc=$?; echo "pmc lds rc=$rc"; if [ "$rc" -ge 124 ]; then exit $rc; fi
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU --kernel-trace -d gpurun_out/pmc_valu -o valu --output-format csv -- python3 bench.py $B > gpurun_out/pmc_valu.log 2>&1
rc=$?; echo "pmc valu rc=$rc"; if [ "$rc" -ge 124 ]; then exit $rc; fi
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_lds", "pmc_valu"):
    rows = []
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
    for r in rows:
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); cnt[k] += 1
    for k in sorted(agg, key=lambda k: -max(agg[k].values()))[:8]:
        n = cnt[k]
        print(f"{k:40s} n={n:3d} " + " ".join(f"{c[3:]}={v / n:.3g}" for c, v in sorted(agg[k].items())))
PY
